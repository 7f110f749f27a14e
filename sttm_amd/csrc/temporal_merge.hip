// Temporal stage of STTM on gfx950.
//
//   k_pairs        candidate pairs + cosine filter            (quadtree_temporal_merger.py:8-73 of the reference)
//                  and, in the LAST pair workgroup of every root-cell column to finish, that column's label stage
//   k_slow_filter  slow_ver: per frame pair, similarity sort + adjacent-duplicate removal                (:75-121)
//   column label stage  label propagation per root-cell column                                          (:223-269)
//                  probe -> grid barrier -> exact replay -> group sizes -> per-frame survivor counts -> N' word.
//                  Runs inside k_pairs (folded), or as k_col_labels<FUSED> (one launch) /
//                  <PROBE> + <FINAL> (two launches, no co-residency assumption)
//   k_group_mean   ranks the survivors of its frame, finds each survivor's members by scanning its column's labels,
//                  per-survivor ascending-order accumulation and mean                                     (:123-171);
//                  its first workgroup publishes N' to the host
//
// All of them address nodes by their ORIGIN ROW  t*H*W + y1*W + x1  in the scratch matrix S written by the
// spatial kernel (1x1 nodes stay in x).  Origin rows are ordered exactly like the reference's sorted node
// indices, so min-label propagation over origin rows is the same computation as over node indices.
//
// Structure that makes this cheap: a node lies inside exactly one root cell and root cells are the same in
// every frame, so (a) candidate pairs never cross root cells -- one workgroup per (frame pair, root cell)
// enumerates <= 16x16 box tests instead of the reference's dense [T-1, M, M, 4] tensor -- and (b) the label
// graph splits into R independent columns (one per root cell, T frames deep).  Only nodes that have a kept edge
// can ever change their label, so a column works on its ACTIVE nodes only (order-preserving compact ids through a
// bitmask over the column's slots): a few hundred ids for a 128-frame clip, which fit the LDS of a pair workgroup.
// The reference's loop is synchronous and stops at the first iteration where ALL labels are idempotent (quirk Q2:
// that is not connected components), so the columns must all run the same number of iterations: every column
// probes to its fixed point and records after which iterations it was idempotent; the global count K is the first
// iteration at which all columns were; a column whose fixed point came later replays exactly K iterations.
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "sttm_kernels.h"
#include "sttm_pairs.inc"

namespace sttm {

__device__ __forceinline__ int ld_agent(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int32_t* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// Columns.  A column = root cell (I, J) over all T frames.  Local slot of leaf (t, y, x) inside the column:
//   s = t*A + (y - Y1)*aw + (x - X1),   A = area of the root cell in leaves; monotone in the origin row.
// ---------------------------------------------------------------------------------------------------
struct Column {
    int Y1, X1, ah, aw, A, slots, base;   // base = T * (leaves of the root cells before this one)
    unsigned mA, maw;                     // ceil(2^32 / A), ceil(2^32 / aw): exact n / d while n * d <= 2^32
    bool fast;                            // slots * A <= 2^32: the multiply-high division by A is exact for every slot
};
__device__ __forceinline__ void root_extent(const LevelDims& g, int I, int J, int& y1, int& y2, int& x1, int& x2) {
    int lo_i = I, hi_i = I, lo_j = J, hi_j = J;
    for (int m = 0; m < g.n_level - 1; ++m) {
        lo_i = child_start(lo_i, g.h[m + 1]);
        hi_i = child_start(hi_i, g.h[m + 1]) + child_count(hi_i, g.h[m + 1]) - 1;
        lo_j = child_start(lo_j, g.w[m + 1]);
        hi_j = child_start(hi_j, g.w[m + 1]) + child_count(hi_j, g.w[m + 1]) - 1;
    }
    y1 = lo_i; y2 = hi_i + 1; x1 = lo_j; x2 = hi_j + 1;
}
__device__ __forceinline__ Column make_column(const TemporalArgs& a, const LevelDims& g, int r) {
    const int I = r / g.w[0], J = r % g.w[0];
    int y1, y2, x1, x2;
    root_extent(g, I, J, y1, y2, x1, x2);
    Column c;
    c.Y1 = y1; c.X1 = x1; c.ah = y2 - y1; c.aw = x2 - x1; c.A = c.ah * c.aw; c.slots = a.T * c.A;
    // leaves owned by root cells 0..r-1: full root rows above + cells to the left in this root row
    c.base = a.T * (y1 * a.W + (y2 - y1) * x1);
    // ceil(2^32 / d) for d >= 2 with 32-bit arithmetic (d == 1 is special-cased by the users)
    c.mA = 0xffffffffu / (unsigned)c.A + 1u;
    c.maw = 0xffffffffu / (unsigned)c.aw + 1u;
    c.fast = (unsigned long long)c.slots * (unsigned long long)c.A <= (1ull << 32);
    return c;
}
// the same from the packed geometry word the spatial kernel left for every leaf position (no level-table walk)
__device__ __forceinline__ Column column_from_geo(const TemporalArgs& a, uint32_t gw) {
    Column c;
    c.Y1 = gw & 255; c.X1 = (gw >> 8) & 255; c.aw = (gw >> 16) & 255; c.ah = gw >> 24;
    c.A = c.ah * c.aw; c.slots = a.T * c.A;
    c.base = a.T * (c.Y1 * a.W + c.ah * c.X1);
    c.mA = 0xffffffffu / (unsigned)c.A + 1u;
    c.maw = 0xffffffffu / (unsigned)c.aw + 1u;
    c.fast = (unsigned long long)c.slots * (unsigned long long)c.A <= (1ull << 32);
    return c;
}
// root cell of leaf (y, x): walk the parents up from the leaf level
__device__ __forceinline__ int root_cell_of(const LevelDims& g, int y, int x) {
    int i = y, j = x;
    for (int l = g.n_level - 1; l >= 1; --l) { i = parent_of(i, g.h[l]); j = parent_of(j, g.w[l]); }
    return i * g.w[0] + j;
}
__device__ __forceinline__ int slot_frame(const Column& c, int s) {
    return c.A == 1 ? s : (c.fast ? (int)__umulhi((unsigned)s, c.mA) : s / c.A);
}
__device__ __forceinline__ int slot_to_row(const TemporalArgs& a, const Column& c, int s) {
    const int t = slot_frame(c, s), q = s - t * c.A;
    const int ly = c.aw == 1 ? q : (int)__umulhi((unsigned)q, c.maw), lx = q - ly * c.aw;       // q * aw < A * A <= 2^32
    return t * a.H * a.W + (c.Y1 + ly) * a.W + (c.X1 + lx);
}

// ---------------------------------------------------------------------------------------------------
// Block-wide exclusive scan of one int per thread; returns the exclusive prefix, *total gets the sum.
// ---------------------------------------------------------------------------------------------------
template <bool LDS_ONLY = false>
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds_wave /*[16]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds_wave[wave] = inc;
    if constexpr (LDS_ONLY) lds_barrier(); else __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nwave; ++w) {
        const int s = lds_wave[w];
        if (w < wave) base += s;
        tot += s;
    }
    if constexpr (LDS_ONLY) lds_barrier(); else __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------------------------------------------
// Column label stage.
// ---------------------------------------------------------------------------------------------------
constexpr int kColThreads = 1024;
constexpr int kMaxProbeIters = 62;
enum { COL_PROBE = 0, COL_FINAL = 1, COL_FUSED = 2 };

// Column arrays live in LDS (GMEM == false) or, for columns too large for it, in a per-column slice of global scratch
// (GMEM == true).  In global memory every access is agent scope (L2-served), so phases separated by __syncthreads()
// see each other's plain writes and atomics alike.
template <bool GMEM> __device__ __forceinline__ int cld(const int* p) {
    if constexpr (GMEM) return ld_agent(p); else return *p;
}
template <bool GMEM> __device__ __forceinline__ void cst(int* p, int v) {
    if constexpr (GMEM) st_agent(p, v); else *p = v;
}
template <bool GMEM> __device__ __forceinline__ void camin(int* p, int v) {
    if constexpr (GMEM) __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicMin(p, v);
}
template <bool GMEM> __device__ __forceinline__ int caadd(int* p, int v) {
    if constexpr (GMEM) return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return atomicAdd(p, v);
}
template <bool GMEM> __device__ __forceinline__ void caor(int* p, int v) {
    if constexpr (GMEM) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicOr(p, v);
}
// Barrier between the phases of a column.  LDS-resident columns exchange everything through LDS, so their barriers
// must not wait for the wave's outstanding global stores (label / group-size tables: consumed by the NEXT kernel) --
// __syncthreads() would drain them every time.  Columns spilled to global scratch need the full barrier.
template <bool GMEM> __device__ __forceinline__ void col_sync() {
    if constexpr (GMEM) __syncthreads(); else lds_barrier();
}

// Working set of one column.  n = active nodes (nodes with at least one kept edge), compact ids 0..n-1 in slot order.
struct ColArrays {
    int* rep;        // [cap] labels (compact ids)            | before the compaction: slot of each raw edge's earlier node
    int* rep2;       // [cap] scatter target / group sizes    | ... and of its later node
    int* cslot;      // [cap] compact id -> column slot
    int* edges;      // [cap] (dst << 16 | src) compact ids   (global scratch: [2 * cap], dst then src)
    int* bits;       // [W] one bit per column slot: the node starting there is active
    int* wpre;       // [W] active nodes before word w
    int* dec;        // [T] nodes of frame t merged away (non-representatives)
    int cap;         // room for active nodes and for kept edges
};
struct ColShared {   // small per-workgroup scratch, carved from dynamic LDS
    unsigned long long wmask[16];
    int wsum[16];
    int part[4][16];
    int flags[4];    // two (changed, not-idempotent) pairs, used by alternate iterations
    int ecount;
    int last;        // k_pairs: this workgroup is the column's last arriver
    int ok;          // grid barrier passed
    int kfast;       // K as read off the barrier word (0: not decided there)
    // dense form (round 4): one flag per iteration (never reset: no barrier between reading a flag and the next iteration), the
    // totals of the closing reduction and the count of waves that have added theirs
    int ch[64], bd[64];
    int tot[4];
    int arrived;
};

template <bool GMEM> __device__ __forceinline__ void edge_get(const int* edges, int e, int& d, int& s) {
    if constexpr (GMEM) { d = ld_agent(edges + 2 * e); s = ld_agent(edges + 2 * e + 1); }
    else { const unsigned pr = (unsigned)edges[e]; d = pr >> 16; s = pr & 0xffffu; }
}
template <bool GMEM> __device__ __forceinline__ void edge_put(int* edges, int e, int d, int s) {
    if constexpr (GMEM) { st_agent(edges + 2 * e, d); st_agent(edges + 2 * e + 1, s); }
    else edges[e] = (int)(((unsigned)d << 16) | (unsigned)s);
}

// One synchronous iteration of get_merge_dst_idx_safe (quadtree_temporal_merger.py:252-265) on compact ids.
//   rep  : labels before (read only during the edge pass)       rep2 : copy of rep that receives the amin scatter
// after the call rep == rep2 == new labels.  flags[0] = some label changed, flags[1] = not idempotent.
template <bool GMEM>
__device__ __forceinline__ void column_iteration(int* rep, int* rep2, const int* edges, int E, int n, int* flags) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { flags[0] = 0; flags[1] = 0; }
    for (int e = tid; e < E; e += nt) {
        int d, s;
        edge_get<GMEM>(edges, e, d, s);
        const int rd = cld<GMEM>(rep + d), rs = cld<GMEM>(rep + s);
        const int m = rd < rs ? rd : rs;
        camin<GMEM>(rep2 + d, m);
        camin<GMEM>(rep2 + s, m);
    }
    col_sync<GMEM>();
    int changed = 0;
    for (int i = tid; i < n; i += nt) {
        const int v = cld<GMEM>(rep2 + cld<GMEM>(rep2 + i));
        if (v != cld<GMEM>(rep + i)) changed = 1;
        cst<GMEM>(rep + i, v);
    }
    if (changed) flags[0] = 1;
    col_sync<GMEM>();
    int bad = 0;
    for (int i = tid; i < n; i += nt) {
        const int v = cld<GMEM>(rep + i);
        cst<GMEM>(rep2 + i, v);
        if (cld<GMEM>(rep + v) != v) bad = 1;
    }
    if (bad) flags[1] = 1;
    col_sync<GMEM>();
}

// Grid-wide barrier among the column workgroups (all resident: see labels_can_fuse / labels_can_fold) that also carries the
// iteration agreement.  The barrier word holds six 10-bit fields: [0] arrivals, [1 + k] columns whose labels are idempotent
// after iteration k + 1 (k < 5).  Every column adds its whole contribution with ONE atomic, so the poll that sees R arrivals
// sees complete counts: K = the first k + 1 whose field equals R -- without the separate read of every column's history
// (*k_lds = 0 when no field qualifies or R > 1023: the caller then reads the histories).  The spin is bounded: on a timeout
// false is returned -- the caller skips the rest of its column instead of continuing with partial data and reports the
// overflow through the N' word (the Python wrapper then fails loudly).
__device__ __forceinline__ bool grid_barrier(unsigned long long* word, int target, unsigned long long history, int* ok_lds, int* k_lds) {
    // Everything that crosses this barrier is written with agent-scope (write-through, sc1) stores or atomics and read
    // with agent-scope loads, so no release/acquire cache maintenance is needed: every wave drains its stores, one lane
    // arrives and polls the word.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool packed = target <= 1023;
        unsigned long long mine = 1ull;
        if (packed) {
#pragma unroll
            for (int k = 0; k < 5; ++k) mine += ((history >> k) & 1ull) << (10 * (k + 1));
        }
        __hip_atomic_fetch_add(word, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();                       // 100 MHz
        int ok = 1;
        unsigned long long w;
        const unsigned long long arrivals_mask = packed ? 0x3ffull : ~0ull;
        while (((w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & arrivals_mask) < (unsigned long long)target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ll) { ok = 0; break; }    // 2 s
        }
        int kf = 0;
        if (ok && packed) {
#pragma unroll
            for (int k = 4; k >= 0; --k)
                if ((int)((w >> (10 * (k + 1))) & 0x3ffull) == target) kf = k + 1;
        }
        *ok_lds = ok;
        *k_lds = kf;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// N' (and the overflow flag) go straight into pinned host memory (first workgroup of k_group_mean): the caller learns N'
// while that kernel still runs.
// The bookkeeping slots (nodes, candidates, edges, iterations) stay device-side: they are diagnostics, added with
// fire-and-forget atomics that nobody waits for, and are complete once the stream has drained.
__device__ __forceinline__ void publish_counts(const TemporalArgs& a, int n_out, int overflow) {
    a.counts[STTM_CNT_OUT] = n_out;
    if (overflow) atomicAdd(a.counts + STTM_CNT_OVERFLOW, overflow);
    if (a.counts_host) {
        __hip_atomic_store(a.counts_host + STTM_CNT_OUT, n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.counts_host + STTM_CNT_OVERFLOW, overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.counts_host + STTM_CNT_SLOTS - 1, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define STTM_LBL_TICK(n) STTM_DEV_TICK(a.dev, lbl_ticks, r == a.dev.lbl_col, n)

// The label stage of column r.  Returns false (before any global side effect) when the column does not fit `arr.cap`;
// the caller then repeats it on global scratch.
//   MODE COL_PROBE : iterate to the fixed point, publish the idempotency history, return
//        COL_FINAL : read every column's history, replay exactly K iterations, write the results
//        COL_FUSED : probe -> grid barrier -> (replay only if K is smaller than this column's own count) -> results
// Results: lab_row / gcnt of the column's active nodes (the spatial kernel wrote the singleton defaults), the column's
// survivors added to frame_cnt[t], bookkeeping counters, and the column's share of N' (+ overflow events) in one 64-bit word.
//   DENSE (LDS only, columns of at most kDenseSlots slots -- the 128-frame headline has 2048; measured slower than the compact form at 8192): ids are the column's slots
//        themselves.  The bitmask only says which slots take part (results are written for those); the prefix over the bit
//        counts, the id table and the raw-pair staging with its conversion pass are gone (round 3: 1.8 of the label kernel's
//        12 us went into the compaction of ~500 ids that fit LDS uncompacted).  Slots without a node keep label == slot.
constexpr int kDenseSlots = 3072;       // (round 4, same-box A/B: slot-indexed better at 2048 / 2880 slots -- 17.1 vs 18.0, 19.6 vs 20.8 us --,
                                        //  compact ids better from 4096 on: 21.3 vs 21.5 us at T=256, 28.1 vs 30.4 us on the 20 x 36 grids,
                                        //  whose 4096-slot edge columns were the slowest of the launch in the slot-indexed form)
template <bool GMEM, int MODE, bool DENSE = false>
__device__ __forceinline__ bool column_labels(const TemporalArgs& a, int r, const Column& col, const ColArrays& arr, ColShared* sh) {
    static_assert(!(GMEM && DENSE), "the dense form lives in LDS");
    const int R = a.R;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
    const int slots = col.slots, W = (slots + 31) >> 5;
    const bool temporal = a.temporal_on && a.T > 1;
    int* rep = arr.rep; int* rep2 = arr.rep2; int* cslot = arr.cslot; int* edges = arr.edges;
    int* bits = arr.bits; int* wpre = arr.wpre; int* dec = arr.dec;
    int* rawd = arr.rep; int* raws = arr.rep2;
    const int nf = a.T - 1;
    int E = 0, nact = 0, K = 0;
    int cand_pre = 0, ovf = 0;
    STTM_LBL_TICK(0);
    // Loads that are consumed much later are issued first: the node counts of this column's (frame, root cell) lists, the
    // candidate counts (bookkeeping), the sticky overflow flag of the pair kernel.
    const unsigned head_pre = tid < a.T ? (unsigned)a.rc_list[(int64_t)(tid * R + r) * a.rc_stride] : 0u;
    if (temporal)
        for (int t = tid; t < nf; t += nt) cand_pre += ld_agent(a.cand_cnt + (int64_t)r * nf + t);
    if (tid == 0) ovf = ld_agent(a.bar + 1);
    for (int t = tid; t < a.T; t += nt) cst<GMEM>(dec + t, 0);
    if (temporal) {
        // ---- this column's kept edges -> raw (slot, slot) pairs + the bitmask of active slots ------------------------------
        const int cap = a.ecap;
        const int32_t* elist = a.edges + (int64_t)r * nf * cap;
        const int32_t* ecnt = a.edge_cnt + (int64_t)r * nf;
        for (int w = tid; w < W; w += nt) cst<GMEM>(bits + w, 0);
        if (tid == 0) sh->ecount = 0;
        if (tid < 64) { sh->ch[tid] = 0; sh->bd[tid] = 0; }
        col_sync<GMEM>();
        auto take = [&](int t, int packed, int pos) {
            const unsigned w = (unsigned)packed;
            const int sd = t * col.A + (int)(w >> 16), ss = (t + 1) * col.A + (int)(w & 0xffffu);
            if constexpr (DENSE) { if (pos < arr.cap) edge_put<GMEM>(edges, pos, sd, ss); }
            else if (pos < arr.cap) { cst<GMEM>(rawd + pos, sd); cst<GMEM>(raws + pos, ss); }
            caor<GMEM>(bits + (sd >> 5), (int)(1u << (sd & 31)));
            caor<GMEM>(bits + (ss >> 5), (int)(1u << (ss & 31)));
        };
        // ONE round trip: the length of every pair's list together with its first HEAD entries (a list holds two edges on
        // average; entries past the length are stale).  The thread that reads entry 0 of a longer list walks the rest.
        // HEAD entries of every list are prefetched: 16 for root cells of up to 16 leaves (a list holds two edges on average there),
        // 64 for larger cells -- on the 20 x 36 / 18 x 26 grids (root cells of 8 x 8 leaves, heavy merging) many lists have 20 - 60
        // entries and the walk of their tails, one dependent round trip per entry, was 7 - 11 us of the busiest column (round 4).
        auto gather = [&](auto per_c, const int hl) {
        constexpr int PER = decltype(per_c)::value;
        const int HEAD = 1 << hl;
        const int total = nf << hl;
        for (int j0 = 0; j0 < total; j0 += PER * nt) {
            int val[PER], tt[PER], cn[PER];
            bool ok[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = j0 + k * nt + tid;
                const bool in = j < total;
                const int t = j >> hl, e = j & (HEAD - 1);
                val[k] = (in && e < cap) ? ld_agent(elist + (int64_t)t * cap + e) : 0;
                cn[k] = in ? ld_agent(ecnt + t) : 0;
                ok[k] = in && e < cn[k];
                tt[k] = t;
            }
            int nv = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) nv += ok[k] ? 1 : 0;
            // compaction with ONE LDS atomic per wave and round (a same-address atomic per kept edge serialises)
            int incl = nv;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            int base = 0;
            if (lane == 63 && incl) base = atomicAdd(&sh->ecount, incl);
            base = __shfl(base, 63, 64);
            int pos = base + incl - nv;
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (ok[k]) take(tt[k], val[k], pos++);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = j0 + k * nt + tid;
                if (j < total && (j & (HEAD - 1)) == 0 && cn[k] > HEAD)
                    for (int e = HEAD; e < cn[k] && e < cap; ++e)
                        take(tt[k], ld_agent(elist + (int64_t)tt[k] * cap + e), atomicAdd(&sh->ecount, 1));
            }
        }
        };
        if (cap > 32) gather(std::integral_constant<int, 8>{}, 6);
        else gather(std::integral_constant<int, 2>{}, 4);
        col_sync<GMEM>();
        E = sh->ecount;
        STTM_LBL_TICK(1);
        if constexpr (DENSE) {
            if (E > arr.cap) return false;
            nact = slots;
            for (int s2 = tid; s2 < slots; s2 += nt) { rep[s2] = s2; rep2[s2] = s2; }
            col_sync<GMEM>();
        } else {
        // ---- order-preserving compact ids: prefix of the bit counts ------------------------------------------------------
        {
            const int per = (W + nt - 1) / nt;
            const int lo = tid * per < W ? tid * per : W, hi = lo + per < W ? lo + per : W;
            int mine = 0;
            for (int w = lo; w < hi; ++w) mine += __popc((unsigned)cld<GMEM>(bits + w));
            int off = block_exclusive_scan<!GMEM>(mine, sh->wsum, &nact);
            for (int w = lo; w < hi; ++w) { cst<GMEM>(wpre + w, off); off += __popc((unsigned)cld<GMEM>(bits + w)); }
        }
        if (E > arr.cap || nact > arr.cap) return false;          // uniform: the caller falls back to global scratch
        col_sync<GMEM>();
        auto compact = [&](int s) {
            const unsigned wbits = (unsigned)cld<GMEM>(bits + (s >> 5));
            return cld<GMEM>(wpre + (s >> 5)) + __popc(wbits & ((1u << (s & 31)) - 1u));
        };
        // raw pairs sit in rep / rep2: all of them are converted before those arrays are initialised
        for (int e = tid; e < E; e += nt)
            edge_put<GMEM>(edges, e, compact(cld<GMEM>(rawd + e)), compact(cld<GMEM>(raws + e)));
        col_sync<GMEM>();
        for (int s = tid; s < slots; s += nt) {
            const unsigned wbits = (unsigned)cld<GMEM>(bits + (s >> 5));
            if ((wbits >> (s & 31)) & 1u) {
                const int c = cld<GMEM>(wpre + (s >> 5)) + __popc(wbits & ((1u << (s & 31)) - 1u));
                cst<GMEM>(cslot + c, s);
                cst<GMEM>(rep + c, c);
                cst<GMEM>(rep2 + c, c);
            }
        }
        col_sync<GMEM>();
        }
        STTM_LBL_TICK(2);
    }

    int probe_iters = 0;
    unsigned long long history = ~0ull;          // this column's idempotency history (all ones: a column without edges never moves)
    if (MODE != COL_FINAL && temporal) {
        // ---- PROBE: iterate to the fixed point, remember after which iterations the labels were idempotent --
        unsigned long long mask = 0ull;
        int it = 0;
        bool overflow = false;
        bool two_barrier = false;
        if constexpr (!GMEM && !DENSE) two_barrier = 2 * nact <= arr.cap && E > 0;      // uniform: room for a third label array
        if (two_barrier) {
            // the dense form's two-barrier iteration (column_labels_dense) on compact ids: the next scatter target lives in the upper
            // half of rep2 (the active nodes of a column that takes this path are a fraction of its slots), the idempotency check of
            // iteration k rides in the edge pass of iteration k + 1, the copy pass is gone
            int* Sc = rep2; int* Sn = rep2 + arr.cap / 2;
            while (true) {
                for (int e = tid; e < E; e += nt) {
                    int d, s2;
                    edge_get<false>(edges, e, d, s2);
                    const int rd = rep[d], rs = rep[s2];
                    const int m = rd < rs ? rd : rs;
                    atomicMin(Sc + d, m);
                    atomicMin(Sc + s2, m);
                }
                if (it > 0) {
                    int bad = 0;
                    for (int i = tid; i < nact; i += nt) {
                        const int v = rep[i];
                        if (rep[v] != v) bad = 1;
                    }
                    if (bad) sh->bd[it - 1] = 1;
                }
                lds_barrier();
                int changed = 0;
                for (int i = tid; i < nact; i += nt) {
                    const int v = Sc[Sc[i]];
                    if (v != rep[i]) changed = 1;
                    rep[i] = v;
                    Sn[i] = v;
                }
                if (changed) sh->ch[it] = 1;
                lds_barrier();
                if (it > 0 && !sh->bd[it - 1]) mask |= 1ull << (it - 1);
                const int chg = sh->ch[it];
                ++it;
                int* tmp = Sc; Sc = Sn; Sn = tmp;
                if (!chg) break;
                if (it >= kMaxProbeIters) { overflow = true; break; }
            }
        } else
        while (true) {
            // alternate flag pairs: iteration k+1 resets the OTHER pair, so no barrier is needed between reading this
            // iteration's flags and starting the next one (the pair is reused two iterations, i.e. >= 3 barriers, later)
            int* fl = sh->flags + 2 * (it & 1);
            column_iteration<GMEM>(rep, rep2, edges, E, nact, fl);
            const int changed = fl[0], bad = fl[1];
            if (!bad) mask |= 1ull << it;
            ++it;
            if (!changed) break;                 // fixed point: stable => idempotent from here on
            if (it >= kMaxProbeIters) { overflow = true; break; }
        }
        mask |= ~0ull << (it - 1);               // the labels no longer move: every later iteration is idempotent
        history = mask;
        if (tid == 0) {
            __hip_atomic_store(a.col_mask + r, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (overflow) ovf += 1;
        }
        probe_iters = it;
    }
    if constexpr (MODE == COL_PROBE) return true;
    STTM_LBL_TICK(3);
    bool alive = true;
    if constexpr (MODE == COL_FUSED) {
        alive = grid_barrier(reinterpret_cast<unsigned long long*>(a.bar + 4), R, history, &sh->ok, &sh->kfast);
        if (!alive) {
            ovf += 1;
            // distinct from list overflows (sticky bit 1 of the flag word): the host retries on the two-launch path
            if (tid == 0) __hip_atomic_fetch_or(a.bar + 1, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    STTM_LBL_TICK(4);
    int nodes = 0, leafnodes = 0, survivors = 0;
    if (alive) {
        // ---- FINAL: K = first iteration after which EVERY column is idempotent; labels after exactly K iterations --
        if (temporal) {
            const int kfast = MODE == COL_FUSED ? sh->kfast : 0;     // the barrier word usually decides K
            if (kfast > 0) {
                K = kfast;
            } else {
                unsigned long long m = ~0ull;
                for (int c = tid; c < R; c += nt) m &= __hip_atomic_load(a.col_mask + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) m &= __shfl_xor(m, d, 64);
                if (lane == 0) sh->wmask[wave] = m;
                col_sync<GMEM>();
                unsigned long long all = ~0ull;
                for (int w = 0; w < nwave; ++w) all &= sh->wmask[w];
                K = all ? __ffsll((long long)all) : kMaxProbeIters;      // lowest set bit index + 1
            }
            // fused: this column already sits at its fixed point, reached after probe_iters - 1 iterations; that is the
            // answer whenever K is at least that.  Otherwise (and in the two-kernel path) replay exactly K iterations.
            if (MODE == COL_FINAL || K < probe_iters - 1) {
                if (MODE == COL_FUSED) {
                    for (int i = tid; i < nact; i += nt) { cst<GMEM>(rep + i, i); cst<GMEM>(rep2 + i, i); }
                    col_sync<GMEM>();
                }
                for (int it = 0; it < K; ++it) column_iteration<GMEM>(rep, rep2, edges, E, nact, sh->flags);
            }
            STTM_LBL_TICK(5);
            // group sizes of the representatives (rep2 is free now)
            for (int i = tid; i < nact; i += nt) cst<GMEM>(rep2 + i, 0);
            col_sync<GMEM>();
            for (int i = tid; i < nact; i += nt) caadd<GMEM>(rep2 + cld<GMEM>(rep + i), 1);
            col_sync<GMEM>();
            // results of the active nodes (plain stores: the consumer is the next kernel); merged-away nodes per frame
            for (int i = tid; i < nact; i += nt) {
                if constexpr (DENSE) {
                    if (!(((unsigned)bits[i >> 5] >> (i & 31)) & 1u)) continue;       // no node with a kept edge starts at this slot
                }
                const int s = DENSE ? i : cld<GMEM>(cslot + i), rr = cld<GMEM>(rep + i);
                const int row = slot_to_row(a, col, s);
                if (rr == i) {
                    a.gcnt[row] = cld<GMEM>(rep2 + i);
                } else {
                    a.lab_row[row] = slot_to_row(a, col, DENSE ? rr : cld<GMEM>(cslot + rr));
                    a.gcnt[row] = 0;
                    caadd<GMEM>(dec + slot_frame(col, s), 1);
                }
            }
            col_sync<GMEM>();
            STTM_LBL_TICK(6);
        }
        // survivors per frame -> frame_cnt (one atomic per (frame, column)); the group-mean kernel turns them into row
        // prefixes.  The spatial kernel left the node count (and the 1x1 count) of every (frame, root cell) in its list head.
        for (int t = tid; t < a.T; t += nt) {
            const unsigned head = t == tid ? head_pre : (unsigned)a.rc_list[(int64_t)(t * R + r) * a.rc_stride];
            const int n_t = (int)(head & 0xffffu), c = n_t - cld<GMEM>(dec + t);
            nodes += n_t; leafnodes += (int)(head >> 16); survivors += c;
            if (c) atomicAdd(a.frame_cnt + t, c);
        }
        STTM_LBL_TICK(7);
    }
    // ---- N' and the bookkeeping counters.  The per-wave partials meet in LDS; thread 0 adds (survivors << 24 | 1) -- plus
    // this column's overflow events in the top byte -- to ONE 64-bit word (R < 2^24 columns, N' < 2^31) with a fire-and-forget
    // atomic, like the diagnostic counters: nobody in this kernel waits for a returned value or drains stores (round 2 had the
    // last column to arrive publish N' itself: a returning atomic plus three host stores at the very end of the 16-workgroup
    // critical path, 1.2 us of kernel time).  The word is complete at the kernel boundary; the first workgroup of the group-mean
    // kernel forwards it to the host.
    int cand = temporal ? cand_pre : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nodes += __shfl_xor(nodes, d, 64); leafnodes += __shfl_xor(leafnodes, d, 64);
        survivors += __shfl_xor(survivors, d, 64); cand += __shfl_xor(cand, d, 64);
    }
    if (lane == 0) { sh->part[0][wave] = nodes; sh->part[1][wave] = leafnodes; sh->part[2][wave] = survivors; sh->part[3][wave] = cand; }
    col_sync<GMEM>();
    // the first wave adds the per-wave partials up: lane (k, w) fetches one, four butterfly steps over w (thread 0 alone walked
    // the 64 LDS words one dependent read after the other: ~1 us at the very end of every column)
    int tot[4] = {0, 0, 0, 0};
    if (wave == 0) {
        const int k = lane >> 4, w = lane & 15;
        int v = w < nwave ? sh->part[k][w] : 0;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tot[kk] = __shfl(v, kk * 16, 64);
    }
    if (tid == 0) {
        unsigned long long* word = reinterpret_cast<unsigned long long*>(a.bar + 2);
        const unsigned long long mine = ((unsigned long long)(ovf > 127 ? 127 : ovf) << 56) | ((unsigned long long)(unsigned)tot[2] << 24) | 1ull;
        (void)__hip_atomic_fetch_add(word, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // early publication (ABI v5): this column's survivors straight into pinned host memory -- one aligned 64-bit store, seq in
        // the upper half, nothing to order and nothing returned; the host adds the R words up a kernel boundary before the
        // group-mean kernel's first workgroup could tell it N'
        if (a.early_host) {
            const bool timed_out = MODE == COL_FUSED && !alive;
            const unsigned long long w = ((unsigned long long)(unsigned)a.seq << 32) | (timed_out ? 0x80000000ull : 0ull) |
                                         ((ovf && !timed_out) ? 0x40000000ull : 0ull) | (unsigned long long)((unsigned)tot[2] & 0x0fffffffu);
            __hip_atomic_store(a.early_host + r, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tot[0]) atomicAdd(a.counts + STTM_CNT_NODES, tot[0]);
        if (tot[1]) atomicAdd(a.counts + STTM_CNT_LEAFNODES, tot[1]);
        if (tot[3]) atomicAdd(a.counts + STTM_CNT_CANDIDATES, tot[3]);
        if (E) atomicAdd(a.counts + STTM_CNT_EDGES, E);
        if (r == 0) st_agent(a.counts + STTM_CNT_ITERS, K);
    }
    STTM_LBL_TICK(8);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// The DENSE form of the column label stage, rebuilt around its barriers (round 4).  Same arithmetic as column_labels<false, MODE,
// true> (ids = the column's slots, LDS only); what changed is the number of workgroup barriers between dependent steps -- at
// 1024 threads every one of them is ~0.2 us of a 16-workgroup kernel that is nothing but dependent steps:
//   * the first round of edge-list loads is requested BEFORE the LDS tables are initialised (the init runs under the round trip);
//     labels and scatter target are initialised in that same phase (the old form spent a pass + a barrier on it after the gather);
//   * an iteration is TWO barriers instead of three: the idempotency check of iteration k (read-only on the labels) rides in the
//     edge pass of iteration k + 1, and the copy pass is gone -- the pointer-jump pass writes the next scatter target itself
//     (three arrays: labels, scatter target, next scatter target; the compact-id table of the other form is free here).  The
//     iteration that finds nothing changed needs no check (a stable labelling is idempotent);
//   * the group-size table is cleared before the grid barrier (whose own barriers order it) instead of behind it;
//   * the closing reduction has no barrier: every wave adds its partial sums to LDS totals and the LAST wave to arrive publishes.
// Returns false (no global side effects yet) when the column has more kept edges than `arr.cap`.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ bool column_labels_dense(const TemporalArgs& a, int r, const Column& col, const ColArrays& arr, ColShared* sh) {
    const int R = a.R;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, nwave = nt >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slots = col.slots, W = (slots + 31) >> 5;
    const bool temporal = a.temporal_on && a.T > 1;
    int* L = arr.rep; int* Sa = arr.rep2; int* Sb = arr.cslot; int* edges = arr.edges;
    int* bits = arr.bits; int* dec = arr.dec;
    const int nf = a.T - 1;
    int E = 0, K = 0, cand_pre = 0, ovf = 0;
    STTM_LBL_TICK(0);
    // ---- loads first: list heads (consumed at the very end), candidate counts, overflow flag, the first round of edge lists ----
    const unsigned head_pre = tid < a.T ? (unsigned)a.rc_list[(int64_t)(tid * R + r) * a.rc_stride] : 0u;
    if (temporal)
        for (int t = tid; t < nf; t += nt) cand_pre += ld_agent(a.cand_cnt + (int64_t)r * nf + t);
    if (tid == 0) ovf = ld_agent(a.bar + 1);
    constexpr int HEAD = 16, PER = 2;           // a list holds two edges on average; one longer than HEAD is walked with dependent loads (rare)
    const int cap = a.ecap;
    const int32_t* elist = a.edges + (int64_t)r * nf * cap;
    const int32_t* ecnt = a.edge_cnt + (int64_t)r * nf;
    const int total = temporal ? nf * HEAD : 0;
    int val[PER], tt[PER], cn[PER];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = j0 + k * nt + tid;
            const bool in = j < total;
            const int t = j / HEAD, e = j % HEAD;
            val[k] = (in && e < cap) ? ld_agent(elist + (int64_t)t * cap + e) : 0;
            cn[k] = in ? ld_agent(ecnt + t) : 0;
            tt[k] = t;
        }
    };
    fetch(0);
    // ---- LDS tables (under the round trip) ----------------------------------------------------------------------------------
    for (int t = tid; t < a.T; t += nt) dec[t] = 0;
    for (int w = tid; w < W; w += nt) bits[w] = 0;
    for (int s2 = tid; s2 < slots; s2 += nt) { L[s2] = s2; Sa[s2] = s2; }
    if (tid < 64) { sh->ch[tid] = 0; sh->bd[tid] = 0; }
    if (tid < 4) sh->tot[tid] = 0;
    if (tid == 0) { sh->ecount = 0; sh->arrived = 0; }
    lds_barrier();
    if (temporal) {
        auto take = [&](int t, int packed, int pos) {
            const unsigned w = (unsigned)packed;
            const int sd = t * col.A + (int)(w >> 16), ss = (t + 1) * col.A + (int)(w & 0xffffu);
            if (pos < arr.cap) edge_put<false>(edges, pos, sd, ss);
            atomicOr(bits + (sd >> 5), (int)(1u << (sd & 31)));
            atomicOr(bits + (ss >> 5), (int)(1u << (ss & 31)));
        };
        for (int j0 = 0; j0 < total; j0 += PER * nt) {
            if (j0 > 0) fetch(j0);
            bool ok[PER];
            int nv = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = j0 + k * nt + tid;
                ok[k] = j < total && (j % HEAD) < cn[k];
                nv += ok[k] ? 1 : 0;
            }
            // compaction with ONE LDS atomic per wave and round (a same-address atomic per kept edge serialises)
            int incl = nv;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            int base = 0;
            if (lane == 63 && incl) base = atomicAdd(&sh->ecount, incl);
            base = __shfl(base, 63, 64);
            int pos = base + incl - nv;
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (ok[k]) take(tt[k], val[k], pos++);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = j0 + k * nt + tid;
                if (j < total && j % HEAD == 0 && cn[k] > HEAD)
                    for (int e = HEAD; e < cn[k] && e < cap; ++e)
                        take(tt[k], ld_agent(elist + (int64_t)tt[k] * cap + e), atomicAdd(&sh->ecount, 1));
            }
        }
        lds_barrier();
        E = sh->ecount;
        if (E > arr.cap) return false;
    }
    STTM_LBL_TICK(1);
    STTM_LBL_TICK(2);
    // one synchronous iteration in the old (three-barrier) form: replays only
    auto replay = [&](int k_iters) {
        for (int i = tid; i < slots; i += nt) { L[i] = i; Sa[i] = i; }
        lds_barrier();
        for (int it2 = 0; it2 < k_iters; ++it2) column_iteration<false>(L, Sa, edges, E, slots, sh->flags);
    };
    int probe_iters = 0;
    unsigned long long history = ~0ull;          // a column without edges never moves
    int* gs = Sb;                                // group sizes go to whichever scatter array is stale at the end
    if (MODE != COL_FINAL && temporal && E > 0) {
        // ---- PROBE: to the fixed point, two barriers per iteration --------------------------------------------------------------
        unsigned long long mask = 0ull;
        int it = 0;
        bool overflow = false;
        int* Sc = Sa; int* Sn = Sb;              // scatter target of this iteration (== labels on entry) | of the next one
        while (true) {
            for (int e = tid; e < E; e += nt) {
                int d, s2;
                edge_get<false>(edges, e, d, s2);
                const int rd = L[d], rs = L[s2];
                const int m = rd < rs ? rd : rs;
                atomicMin(Sc + d, m);
                atomicMin(Sc + s2, m);
            }
            if (it > 0) {                        // idempotency of the labels iteration it - 1 left (read-only: rides along)
                int bad = 0;
                for (int i = tid; i < slots; i += nt) {
                    const int v = L[i];
                    if (L[v] != v) bad = 1;
                }
                if (bad) sh->bd[it - 1] = 1;
            }
            lds_barrier();
            int changed = 0;
            for (int i = tid; i < slots; i += nt) {
                const int v = Sc[Sc[i]];
                if (v != L[i]) changed = 1;
                L[i] = v;
                Sn[i] = v;
            }
            if (changed) sh->ch[it] = 1;
            lds_barrier();
            if (it > 0 && !sh->bd[it - 1]) mask |= 1ull << (it - 1);
            const int chg = sh->ch[it];
            ++it;
            int* tmp = Sc; Sc = Sn; Sn = tmp;
            if (!chg) break;                     // fixed point: stable => idempotent from here on
            if (it >= kMaxProbeIters) { overflow = true; break; }
        }
        mask |= ~0ull << (it - 1);               // the last iteration changed nothing: it and every later one are idempotent ...
        // (iteration it - 2's own check rode in iteration it - 1's edge pass)
        history = mask;
        if (tid == 0) {
            __hip_atomic_store(a.col_mask + r, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (overflow) ovf += 1;
        }
        probe_iters = it;
        gs = Sn;                                 // stale; Sc is a copy of the labels
    } else if (MODE != COL_FINAL && temporal && tid == 0) {
        __hip_atomic_store(a.col_mask + r, history, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        probe_iters = 1;
    }
    if (MODE != COL_FINAL && temporal && E == 0) probe_iters = 1;
    if constexpr (MODE == COL_PROBE) return true;
    STTM_LBL_TICK(3);
    for (int i = tid; i < slots; i += nt) gs[i] = 0;            // ordered by the barriers below
    bool alive = true;
    if constexpr (MODE == COL_FUSED) {
        alive = grid_barrier(reinterpret_cast<unsigned long long*>(a.bar + 4), R, history, &sh->ok, &sh->kfast);
        if (!alive) {
            ovf += 1;
            if (tid == 0) __hip_atomic_fetch_or(a.bar + 1, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        lds_barrier();
    }
    STTM_LBL_TICK(4);
    int nodes = 0, leafnodes = 0, survivors = 0;
    if (alive) {
        if (temporal) {
            const int kfast = MODE == COL_FUSED ? sh->kfast : 0;
            if (kfast > 0) {
                K = kfast;
            } else {
                unsigned long long m = ~0ull;
                for (int c = tid; c < R; c += nt) m &= __hip_atomic_load(a.col_mask + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) m &= __shfl_xor(m, d, 64);
                if (lane == 0) sh->wmask[wave] = m;
                lds_barrier();
                unsigned long long all = ~0ull;
                for (int w = 0; w < nwave; ++w) all &= sh->wmask[w];
                K = all ? __ffsll((long long)all) : kMaxProbeIters;
            }
            if (E > 0 && (MODE == COL_FINAL || K < probe_iters - 1)) {
                replay(K);                                   // ends with a barrier; both scatter arrays of the probe are dead now
                gs = Sb;
                for (int i = tid; i < slots; i += nt) gs[i] = 0;
                lds_barrier();
            }
            STTM_LBL_TICK(5);
            if (E > 0) {
                for (int i = tid; i < slots; i += nt) atomicAdd(gs + L[i], 1);
                lds_barrier();
                for (int i = tid; i < slots; i += nt) {
                    if (!(((unsigned)bits[i >> 5] >> (i & 31)) & 1u)) continue;       // no node with a kept edge starts at this slot
                    const int rr = L[i];
                    const int row = slot_to_row(a, col, i);
                    if (rr == i) {
                        a.gcnt[row] = gs[i];
                    } else {
                        a.lab_row[row] = slot_to_row(a, col, rr);
                        a.gcnt[row] = 0;
                        atomicAdd(dec + slot_frame(col, i), 1);
                    }
                }
                lds_barrier();
            }
            STTM_LBL_TICK(6);
        }
        for (int t = tid; t < a.T; t += nt) {
            const unsigned head = t == tid ? head_pre : (unsigned)a.rc_list[(int64_t)(t * R + r) * a.rc_stride];
            const int n_t = (int)(head & 0xffffu), c = n_t - dec[t];
            nodes += n_t; leafnodes += (int)(head >> 16); survivors += c;
            if (c) atomicAdd(a.frame_cnt + t, c);
        }
        STTM_LBL_TICK(7);
    }
    // ---- N' and the bookkeeping counters, without a barrier: the last wave to add its partial sums publishes --------------------
    int cand = temporal ? cand_pre : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nodes += __shfl_xor(nodes, d, 64); leafnodes += __shfl_xor(leafnodes, d, 64);
        survivors += __shfl_xor(survivors, d, 64); cand += __shfl_xor(cand, d, 64);
    }
    if (lane == 0) {
        if (nodes) atomicAdd(&sh->tot[0], nodes);
        if (leafnodes) atomicAdd(&sh->tot[1], leafnodes);
        if (survivors) atomicAdd(&sh->tot[2], survivors);
        if (cand) atomicAdd(&sh->tot[3], cand);
        atomicAdd(&sh->arrived, 1);                              // (the LDS operations of a wave are performed in order)
    }
    if (tid == 0) {                                              // thread 0 carries the overflow events: it publishes
        while (atomicAdd(&sh->arrived, 0) < nwave) __builtin_amdgcn_s_sleep(1);       // LDS only; the other waves are a few instructions away
        const int t0 = atomicAdd(&sh->tot[0], 0), t1 = atomicAdd(&sh->tot[1], 0), t2 = atomicAdd(&sh->tot[2], 0), t3 = atomicAdd(&sh->tot[3], 0);
        unsigned long long* word = reinterpret_cast<unsigned long long*>(a.bar + 2);
        const unsigned long long mine = ((unsigned long long)(ovf > 127 ? 127 : ovf) << 56) | ((unsigned long long)(unsigned)t2 << 24) | 1ull;
        (void)__hip_atomic_fetch_add(word, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.early_host) {
            const bool timed_out = MODE == COL_FUSED && !alive;
            const unsigned long long w = ((unsigned long long)(unsigned)a.seq << 32) | (timed_out ? 0x80000000ull : 0ull) |
                                         ((ovf && !timed_out) ? 0x40000000ull : 0ull) | (unsigned long long)((unsigned)t2 & 0x0fffffffu);
            __hip_atomic_store(a.early_host + r, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (t0) atomicAdd(a.counts + STTM_CNT_NODES, t0);
        if (t1) atomicAdd(a.counts + STTM_CNT_LEAFNODES, t1);
        if (t3) atomicAdd(a.counts + STTM_CNT_CANDIDATES, t3);
        if (E) atomicAdd(a.counts + STTM_CNT_EDGES, E);
        if (r == 0) st_agent(a.counts + STTM_CNT_ITERS, K);
    }
    STTM_LBL_TICK(8);
    return true;
}

// LDS bytes of a column working set with room for `cap` active nodes / edges
__host__ __device__ inline size_t col_lds_bytes(int cap, int slots, int T) {
    const size_t W = ((size_t)slots + 31) / 32;
    size_t b = sizeof(ColShared) + 4 * ((size_t)4 * cap + 2 * W + T);
    return (b + 15) / 16 * 16;
}
__device__ __forceinline__ ColArrays col_arrays_lds(char* smem, int cap, int slots, int T, ColShared** sh) {
    const int W = (slots + 31) >> 5;
    *sh = reinterpret_cast<ColShared*>(smem);
    int* p = reinterpret_cast<int*>(smem + sizeof(ColShared));
    ColArrays A;
    A.rep = p; A.rep2 = p + cap; A.cslot = p + 2 * cap; A.edges = p + 3 * cap; A.bits = p + 4 * cap; A.wpre = A.bits + W; A.dec = A.wpre + W;
    A.cap = cap;
    return A;
}
// global scratch: per column  rep, rep2, cslot [2 * slots each: the raw edge pairs need that much], edges [4 * slots],
// bits, wpre [W each], dec [T]; column r starts at  10 * base + 2 * (base / 32) + r * (T + 18)
size_t colscratch_ints(int T, int H, int W, int R) {
    const size_t N = (size_t)T * H * W;
    return 10 * N + 2 * (N / 32) + (size_t)(R + 1) * (T + 18);
}
__device__ __forceinline__ ColArrays col_arrays_gmem(const TemporalArgs& a, const Column& col, int r) {
    const int W = (col.slots + 31) >> 5;
    int* p = a.colscratch + (size_t)10 * col.base + 2 * ((size_t)col.base / 32) + (size_t)r * (a.T + 18);
    ColArrays A;
    A.cap = 2 * col.slots;
    A.rep = p; A.rep2 = p + A.cap; A.cslot = p + 2 * (size_t)A.cap; A.edges = p + 3 * (size_t)A.cap; A.bits = p + 5 * (size_t)A.cap;
    A.wpre = A.bits + W; A.dec = A.wpre + W;
    return A;
}

// LDS first (if the launch gave this workgroup room for `cap` > 0 ids), global scratch when the column does not fit
template <int MODE>
__device__ __forceinline__ void column_labels_any(const TemporalArgs& a, const LevelDims& g, int r, char* smem, int cap) {
    const Column col = make_column(a, g, r);
    ColShared* sh;
    const ColArrays lds = col_arrays_lds(smem, cap > 0 ? cap : 0, cap > 0 ? col.slots : 0, cap > 0 ? a.T : 0, &sh);
    if (cap > 0 && !a.force_gmem) {
        if (col.slots <= kDenseSlots && col.slots <= cap && a.no_dense != 1) {
            // (false: more kept edges than room -- cannot happen while cap >= slots; the compact form decides)
            if (a.no_dense == 2 ? column_labels<false, MODE, true>(a, r, col, lds, sh) : column_labels_dense<MODE>(a, r, col, lds, sh)) return;
            __syncthreads();
        }
        if (column_labels<false, MODE>(a, r, col, lds, sh)) return;
        __syncthreads();
    }
    column_labels<true, MODE>(a, r, col, col_arrays_gmem(a, col, r), sh);
}

template <int MODE>
__global__ void __launch_bounds__(kColThreads) k_col_labels(const TemporalArgs a0, const BatchPtrs bp, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // The level table is indexed dynamically: it is read through the (constant) kernel argument, while the per-video
    // pointers live in a copy that the compiler keeps in registers.
    const LevelDims& g = a0.dims;
    TemporalArgs a = a0;
    rebase(a, bp, blockIdx.y);
    column_labels_any<MODE>(a, g, blockIdx.x, smem_raw, cap);
}

// Device facts the residency decisions need, cached per DEVICE (a process may drive several: the answer for the device that
// happened to be current at first use is not the answer for the others).
constexpr int kMaxDevices = 64;
static int device_cus() {
    static std::atomic<int> cache[kMaxDevices];          // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    if (dev < kMaxDevices) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 0;
    if (dev < kMaxDevices) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}
static int device_cus_or(int dflt) { const int n = device_cus(); return n > 0 ? n : dflt; }

// ---------------------------------------------------------------------------------------------------
// K2: pairs.  One workgroup per (root cell, run of `pairs_seg` consecutive frame pairs): box tests between the node lists of
// every frame t and t+1 of the run, then one wave per candidate for the C-long dot product (two candidates in flight per
// wave; the candidates of all pairs of the run form one work list, so the waves stay balanced).  A run reads the node rows
// of its inner frames for both of their pairs from the same CU (L1 / L2 hits).  Kept edges go to the pair's own list -- no
// global counters.  With a.fold_labels the workgroup then announces itself on its column's arrival counter, and the last one
// to arrive runs the column's label stage right here, with all of its (up to 1024) threads: the stand-alone label kernel
// (R workgroups, one launch ramp and one tail) disappears.
// ---------------------------------------------------------------------------------------------------
struct PairShared { int last, pad0, pad1, pad2; };

template <typename T, int VEC, int UNRB, bool LEAN>
__device__ __forceinline__ void pairs_body(const TemporalArgs& a0, const BatchPtrs& bp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const LevelDims& g = a0.dims;          // see k_col_labels
    TemporalArgs a = a0;
    rebase(a, bp, blockIdx.y);
#define STTM_K2_TICK(n) STTM_DEV_TICK(a.dev, k2_ticks, blockIdx.x == a.dev.k2_wg, n)
    STTM_K2_TICK(0);
    const int L = a.pairs_seg, nf = a.T - 1, nseg = (nf + L - 1) / L;
    // XCD-aware order of the runs.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2, and the node rows of a
    // frame are read by the run that ends with it and by the run that starts with it: keep kXcdRuns consecutive runs of one
    // root cell on ONE XCD so that the second read hits that L2 instead of going to the fabric again (measured with
    // one-pair runs: 19.2 -> 16.0 us, fetch 76 -> 68 MB).
    constexpr int kXcdRuns = 16;
    const int total_runs = a.R * nseg;
    int q;
    {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        q = ((i / kXcdRuns) * 8 + xcd) * kXcdRuns + i % kXcdRuns;
    }
    if (q >= total_runs) return;
    const int r = q / nseg, sg = q - r * nseg;
    const int t0 = sg * L, np = nf - t0 < L ? nf - t0 : L;           // this workgroup: pairs t0 .. t0 + np - 1
    PairShared* ps = reinterpret_cast<PairShared*>(smem_raw);
    const Column col = make_column(a, g, r);
    const PairGeo geo = {col.Y1, col.X1, col.aw};
    const int tid = threadIdx.x;
    if (tid == 0) ps->last = 0;
    if constexpr (LEAN) {          // no folded label stage, no per-head cosine: the common call
        pair_run<T, VEC, false, false, UNRB, false>(a, geo, r, t0, np, L, smem_raw + sizeof(PairShared));
        return;
    }
    if (a.fold_labels) pair_run<T, VEC, false, true, UNRB>(a, geo, r, t0, np, L, smem_raw + sizeof(PairShared));
    else pair_run<T, VEC, false, false, UNRB>(a, geo, r, t0, np, L, smem_raw + sizeof(PairShared));
    STTM_K2_TICK(2);
    if (a.fold_labels && tid == 0) {
        const int old = __hip_atomic_fetch_add(a.col_arrive + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ps->last = old == nseg - 1 ? 1 : 0;
    }
    STTM_K2_TICK(3);
    if (!a.fold_labels) return;
    __syncthreads();
    if (!ps->last) return;
    __syncthreads();               // ps is about to be overlaid by the label stage's scratch
    // ---- this workgroup completed column r: its label stage runs here, in the LDS the pair phase no longer needs ----------
    column_labels_any<COL_FUSED>(a, g, r, smem_raw, a.fold_cap);
#undef STTM_K2_TICK
}
// any block size up to 1024 (runs of pairs, folded label stage)
template <typename T, int VEC>
__global__ void __launch_bounds__(1024) k_pairs(const TemporalArgs a0, const BatchPtrs bp) { pairs_body<T, VEC, 64, false>(a0, bp); }
// the default shape: one pair per 256-thread workgroup, registers bounded for OCC waves per SIMD
template <typename T, int VEC, int OCC, int UNRB>
__global__ void __launch_bounds__(256, OCC) k_pairs256(const TemporalArgs a0, const BatchPtrs bp) { pairs_body<T, VEC, UNRB, true>(a0, bp); }

static size_t pairs_smem(const TemporalArgs& a) {
    return sizeof(PairShared) + pair_lds_bytes(a.pairs_seg, a.rc_stride, a.ecap);
}

hipError_t launch_pairs(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream) {
    if (a.T < 2) return hipSuccess;
    const int nseg = (a.T - 1 + a.pairs_seg - 1) / a.pairs_seg;
    const int grid = (a.R * nseg + 127) / 128 * 128;                  // whole (8 XCDs x 16 runs) chunks of the XCD-aware order
    size_t smem = pairs_smem(a);
    if (a.fold_labels) {
        const size_t lb = col_lds_bytes(a.fold_cap, a.max_slots, a.T);
        if (lb > smem) smem = lb;
    }
    const int nt = a.pairs_nt;
#define STTM_LAUNCH_PAIRS(TT, VV) hipLaunchKernelGGL((k_pairs<TT, VV>), dim3(grid, n_videos), dim3(nt), smem, stream, a, bp)
#define STTM_LAUNCH_PAIRS256(TT, VV, OCC, UB) hipLaunchKernelGGL((k_pairs256<TT, VV, OCC, UB>), dim3(grid, n_videos), dim3(256), smem, stream, a, bp)
    // default shape (one pair per 256-thread workgroup, whole-vector cosine, stand-alone label stage): the lean kernel --
    // rows through buffer descriptors with scalar bases, registers bounded for 5 waves per SIMD (A/B on MI355X, round 3:
    // fp32 C=1024 one 32-byte pack per row in flight 16.5 -> ~13.5 us; 16-bit rows 64 bytes per row in flight, bf16 C=3584
    // 32.8 -> 27.7 us; variants bounded to 6 / 8 waves spilled and were slower).  pairs_var = 9 selects the general kernel.
    if (nt == 256 && a.pairs_var != 9 && !a.fold_labels && a.n_head == 0) {
        if (a.dtype == STTM_F32) {
            if (a.pair_vec == 8) STTM_LAUNCH_PAIRS256(float, 8, 5, 32); else if (a.pair_vec == 4) STTM_LAUNCH_PAIRS256(float, 4, 5, 32);
            else if (a.pair_vec == 2) STTM_LAUNCH_PAIRS256(float, 2, 5, 32); else STTM_LAUNCH_PAIRS256(float, 1, 5, 32);
        } else if (a.dtype == STTM_BF16) {
            if (a.pair_vec == 8) STTM_LAUNCH_PAIRS256(bf16_t, 8, 5, 64); else if (a.pair_vec == 4) STTM_LAUNCH_PAIRS256(bf16_t, 4, 5, 64); else STTM_LAUNCH_PAIRS256(bf16_t, 2, 5, 64);
        } else {
            if (a.pair_vec == 8) STTM_LAUNCH_PAIRS256(f16_t, 8, 5, 64); else if (a.pair_vec == 4) STTM_LAUNCH_PAIRS256(f16_t, 4, 5, 64); else STTM_LAUNCH_PAIRS256(f16_t, 2, 5, 64);
        }
        return hipGetLastError();
    }
    if (a.dtype == STTM_F32) {
        if (a.pair_vec == 8) STTM_LAUNCH_PAIRS(float, 8); else if (a.pair_vec == 4) STTM_LAUNCH_PAIRS(float, 4); else if (a.pair_vec == 2) STTM_LAUNCH_PAIRS(float, 2); else STTM_LAUNCH_PAIRS(float, 1);
    } else if (a.dtype == STTM_BF16) {
        if (a.pair_vec == 8) STTM_LAUNCH_PAIRS(bf16_t, 8); else if (a.pair_vec == 4) STTM_LAUNCH_PAIRS(bf16_t, 4); else STTM_LAUNCH_PAIRS(bf16_t, 2);
    } else {
        if (a.pair_vec == 8) STTM_LAUNCH_PAIRS(f16_t, 8); else if (a.pair_vec == 4) STTM_LAUNCH_PAIRS(f16_t, 4); else STTM_LAUNCH_PAIRS(f16_t, 2);
    }
#undef STTM_LAUNCH_PAIRS
    return hipGetLastError();
}

// Frame pairs per pair workgroup (a run) and its block size.  Default: one pair per 256-thread workgroup -- the most
// workgroups in flight, measured fastest for the pair phase (T=128 headline: 1 pair x 256 threads 15.3 us; runs of 4 / 8
// pairs with 512 / 1024 threads 18-19 us: a CU with one big workgroup streams its rows at the per-CU rate).  Runs are for
// the folded label stage (about one workgroup per CU, so the whole grid is resident) and for experiments.
void pairs_shape(int T, int R, int fold, int want_seg, int want_nt, int* seg, int* nt) {
    const int nf = T > 1 ? T - 1 : 1;
    int L = want_seg > 0 ? want_seg : 1;
    if (want_seg <= 0 && fold) {
        const int cus = device_cus_or(256);
        L = (int)(((long)nf * R + cus - 1) / cus);
    }
    if (L < 1) L = 1;
    if (L > 64) L = 64;
    if (L > nf) L = nf;
    int n = want_nt > 0 ? want_nt : (L == 1 ? 256 : 128 * L);
    const int lo = want_nt > 0 ? 64 : 256;
    if (n < lo) n = lo;
    if (n > 1024) n = 1024;
    // the threads of a run are split evenly over a power-of-two number of pairs: the block size is a power of two
    int p2 = 64;
    while (p2 * 2 <= n) p2 *= 2;
    *seg = L; *nt = p2;
}

// OPT-IN ("fold_labels" switch; measured no faster than the stand-alone label kernel on the headline -- DESIGN.md -- and prone to
// barrier stalls when several streams run it at once): fold the label stage into the pair kernel when (a) there is a pair kernel and no slow_ver filter between the two, (b) the
// column's active nodes get a useful amount of LDS, and (c) the R * n_videos workgroups that wait for each other at the
// in-kernel grid barrier occupy at most half of the CUs (a 1024-thread workgroup can take a CU for itself), so every other
// workgroup of the launch still finds a CU and runs to completion.
bool labels_can_fold(const TemporalArgs& a, int n_videos, int* cap) {
    if (!a.want_fold || a.slow_ver || !a.temporal_on || a.T < 2) return false;
    const int cus = device_cus();
    if (cus <= 0 || (long long)a.R * n_videos > cus / 2) return false;
    // ... and the whole pair grid runs in ONE round (a workgroup of 512+ threads has a CU to itself): grids of several
    // rounds were seen to stall at the barrier, so they take the stand-alone label kernel
    const int nseg = (a.T - 1 + a.pairs_seg - 1) / a.pairs_seg;
    if ((long long)a.R * nseg * n_videos > (long long)cus * (a.pairs_nt >= 512 ? 1 : 2)) return false;
    const size_t budget = (size_t)a.fold_kb * 1024;
    const size_t fixed = col_lds_bytes(0, a.max_slots, a.T);
    if (fixed + 16 * 256 > budget) return false;
    int c = (int)((budget - fixed) / 16);
    if (c > a.max_slots) c = a.max_slots;
    if (c > 65535) c = 65535;                             // LDS edges pack two compact ids into 32 bits
    *cap = c;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// slow_ver (get_cross_frame_node_pairs_slow, quadtree_temporal_merger.py:75-121): per frame pair, the kept edges of
// ALL root cells are ordered by similarity, descending, and an edge is dropped when its src (later-frame node)
// equals the src of the edge right before it in that order -- adjacent-duplicate removal, not an arg-max per src.
// One workgroup per frame pair: gather -> bitonic sort in LDS -> filter -> rewrite the per-column lists.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_slow_filter(const TemporalArgs a0, const BatchPtrs bp, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TemporalArgs a = a0;
    rebase(a, bp, blockIdx.y);
    float* key = reinterpret_cast<float*>(smem_raw);            // [npad] similarity
    int* ent = reinterpret_cast<int*>(key + npad);               // [npad] packed leaf offsets (dst << 16 | src)
    int* colr = ent + npad;                                       // [npad] root cell
    int* ccnt = colr + npad;                                      // [R] rebuilt list lengths
    __shared__ int n_sh;
    const int t = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nf = a.T - 1, R = a.R, cap = a.ecap;
    if (tid == 0) n_sh = 0;
    for (int r = tid; r < R; r += nt) ccnt[r] = 0;
    __syncthreads();
    for (int j = tid; j < R * cap; j += nt) {
        const int r = j / cap, e = j - r * cap;
        const int64_t cidx = (int64_t)r * nf + t;
        if (e < a.edge_cnt[cidx]) {
            const int pos = atomicAdd(&n_sh, 1);
            key[pos] = a.edge_sim[cidx * cap + e];
            ent[pos] = a.edges[cidx * cap + e];
            colr[pos] = r;
        }
    }
    __syncthreads();
    const int n = n_sh;
    for (int i = n + tid; i < npad; i += nt) { key[i] = -INFINITY; ent[i] = -1; colr[i] = -1; }
    __syncthreads();
    // the gather order is nondeterministic (atomics): make the sort total by breaking similarity ties on
    // (root cell, packed entry), so the result does not depend on arrival order
    auto before = [&](int i, int j) {       // true if element i must come before element j (descending similarity)
        if (key[i] != key[j]) return key[i] > key[j];
        if (colr[i] != colr[j]) return colr[i] < colr[j];
        return ent[i] < ent[j];
    };
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npad; i += nt) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool swap = up ? before(l, i) : before(i, l);
                    if (swap) {
                        const float fk = key[i]; key[i] = key[l]; key[l] = fk;
                        const int fe = ent[i]; ent[i] = ent[l]; ent[l] = fe;
                        const int fc = colr[i]; colr[i] = colr[l]; colr[l] = fc;
                    }
                }
            }
            __syncthreads();
        }
    }
    // drop an edge whose src equals the previous edge's src (same root cell and same src leaf)
    for (int i = tid; i < n; i += nt) {
        const bool dup = i > 0 && colr[i] == colr[i - 1] && (ent[i] & 0xffff) == (ent[i - 1] & 0xffff);
        if (!dup) {
            const int r = colr[i];
            const int pos = atomicAdd(&ccnt[r], 1);
            a.edges[((int64_t)r * nf + t) * cap + pos] = ent[i];
        }
    }
    __syncthreads();
    for (int r = tid; r < R; r += nt) a.edge_cnt[(int64_t)r * nf + t] = ccnt[r];
}

hipError_t launch_slow_filter(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream) {
    if (a.T < 2) return hipSuccess;
    int npad = 2;
    while (npad < a.R * a.ecap) npad <<= 1;                       // worst case: every list full
    const size_t smem = sizeof(int) * ((size_t)3 * npad + a.R);
    if (smem > 150 * 1024) return hipErrorInvalidValue;
    int nthreads = npad < 1024 ? (npad < 64 ? 64 : npad) : 1024;
    hipLaunchKernelGGL(k_slow_filter, dim3(a.T - 1, n_videos), dim3(nthreads), smem, stream, a, bp, npad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Stand-alone label kernels (slow_ver, spatial-only calls, columns beyond the fold budget, R beyond the residency bound)
// ---------------------------------------------------------------------------------------------------
constexpr size_t kColLdsLimit = 160 * 1024 - 2048;

// ids a stand-alone column workgroup gets room for in LDS (0 = straight to global scratch)
static int col_kernel_cap(const TemporalArgs& a) {
    if (a.force_gmem) return 0;
    const size_t fixed = col_lds_bytes(0, a.max_slots, a.T);
    if (fixed + 16 * 64 > kColLdsLimit) return 0;
    size_t c = (kColLdsLimit - fixed) / 16;
    if (c > (size_t)a.max_slots) c = a.max_slots;
    if (c > 65535) c = 65535;
    return (int)c;
}

int col_threads(const TemporalArgs& a) {
    const int cap = a.label_nt;
    int nthreads = 256;
    while (nthreads < cap && nthreads * 2 <= a.max_slots) nthreads *= 2;
    return nthreads;
}

hipError_t launch_col_labels(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, bool probe, hipStream_t stream) {
    const int cap = col_kernel_cap(a);
    const int nthreads = col_threads(a);
    const size_t smem = col_lds_bytes(cap, cap > 0 ? a.max_slots : 0, cap > 0 ? a.T : 0);
    if (probe) {
        if (!(a.temporal_on && a.T > 1)) return hipSuccess;
        hipLaunchKernelGGL((k_col_labels<COL_PROBE>), dim3(a.R, n_videos), dim3(nthreads), smem, stream, a, bp, cap);
    } else {
        hipLaunchKernelGGL((k_col_labels<COL_FINAL>), dim3(a.R, n_videos), dim3(nthreads), smem, stream, a, bp, cap);
    }
    return hipGetLastError();
}

// One launch for probe + final when every column workgroup is certainly resident at once: the global agreement becomes
// an in-kernel grid barrier.  Residency: R workgroups against what the occupancy calculator says the device holds of this
// kernel with this much LDS (it is one workgroup per CU for the large-LDS shapes) -- a partitioned device (CPX mode,
// 32 CUs) or a smaller part simply takes the two-launch path.
bool labels_can_fuse(const TemporalArgs& a, int n_videos, int concurrent_sets) {
    if (a.no_fuse) return false;
    const int cus = device_cus();
    if (cus <= 0) return false;
    const int cap = col_kernel_cap(a);
    const size_t smem = col_lds_bytes(cap, cap > 0 ? a.max_slots : 0, cap > 0 ? a.T : 0);
    // the occupancy query is a driver call: asked once per (device, block size, LDS bytes), not on every merge
    int per_cu = 0;
    {
        static std::mutex mu;
        static std::map<std::tuple<int, int, size_t>, int> memo;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const auto key = std::make_tuple(dev, col_threads(a), smem);
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(key);
        if (it == memo.end()) {
            int v = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_col_labels<COL_FUSED>, col_threads(a), smem) != hipSuccess) v = 0;
            it = memo.emplace(key, v).first;
        }
        per_cu = it->second;
    }
    if (per_cu < 1) return false;
    // The API answers one block per CU too many for kernels with > 80 SGPRs (MI355X_MICROARCH.md, residency): half of what it promises is
    // safe from per_cu = 2 on.  Who else holds CUs: the stage-skewed batch form says how many launch sets can sit in this kernel's grid
    // barrier at once (`concurrent_sets`: they share the budget -- round-5 advisor finding); a lone call knows nothing about the caller's
    // other streams and keeps the quarter it always took.
    if (concurrent_sets > 1) return (long long)a.R * n_videos * concurrent_sets <= (long long)per_cu * cus / 2;
    return (long long)a.R * n_videos <= (long long)per_cu * cus / 4;
}

hipError_t launch_labels_fused(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream) {
    const int cap = col_kernel_cap(a);
    const size_t smem = col_lds_bytes(cap, cap > 0 ? a.max_slots : 0, cap > 0 ? a.T : 0);
    hipLaunchKernelGGL((k_col_labels<COL_FUSED>), dim3(a.R, n_videos), dim3(col_threads(a)), smem, stream, a, bp, cap);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Generic single-workgroup label propagation on an explicit edge list (sttm_merge_dst_idx): global
// memory, agent-scope accesses so phases separated by __syncthreads() see each other's writes.
// ---------------------------------------------------------------------------------------------------
constexpr int kLabelThreads = 1024;

__global__ void __launch_bounds__(kLabelThreads) k_label_edges(const int32_t* pairs, int L, int N, int32_t* rep_out,
                                                               int32_t* rep2, int32_t* emin, int32_t* iters_out) {
    __shared__ int flag;
    const int tid = threadIdx.x, nt = blockDim.x;
    int32_t* cur = rep_out;
    int32_t* nxt = rep2;
    for (int i = tid; i < N; i += nt) st_agent(cur + i, i);
    __syncthreads();
    int iters = 0;
    while (true) {
        for (int e = tid; e < L; e += nt) {
            const int rd = ld_agent(cur + pairs[2 * e]), rs = ld_agent(cur + pairs[2 * e + 1]);
            emin[e] = rd < rs ? rd : rs;
        }
        if (tid == 0) flag = 0;
        __syncthreads();
        for (int e = tid; e < L; e += nt) {
            const int m = emin[e];
            __hip_atomic_fetch_min(cur + pairs[2 * e], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min(cur + pairs[2 * e + 1], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int i = tid; i < N; i += nt) st_agent(nxt + i, ld_agent(cur + ld_agent(cur + i)));
        __syncthreads();
        int bad = 0;
        for (int i = tid; i < N; i += nt) {
            const int v = ld_agent(nxt + i);
            if (ld_agent(nxt + v) != v) bad = 1;
        }
        if (bad) flag = 1;
        __syncthreads();
        const int again = flag;
        __syncthreads();
        int32_t* tmp = cur; cur = nxt; nxt = tmp;
        ++iters;
        if (!again) break;
    }
    if (cur != rep_out) {
        for (int i = tid; i < N; i += nt) st_agent(rep_out + i, ld_agent(cur + i));
    }
    if (tid == 0 && iters_out) *iters_out = iters;
}

hipError_t launch_label_edges(const int32_t* pairs, int L, int N, int32_t* rep_out, int32_t* rep2, int32_t* emin,
                              int32_t* iters_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_label_edges, dim3(1), dim3(kLabelThreads), 0, stream, pairs, L, N, rep_out, rep2, emin, iters_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// K5: group mean.  One wave per output row; members are accumulated in ascending origin order in the
// INPUT dtype (one rounding per add for bf16/fp16, like the reference's index_add_), then divided by
// the member count (or the patch count when weighted_avg).  A survivor with more than one member finds them
// by scanning the labels of its column from its own slot on (64 slots per step, ascending = accumulation order).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
template <typename T> __device__ __forceinline__ float round_to(float f);
template <> __device__ __forceinline__ float round_to<float>(float f) { return f; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float f) { return bf16_bits_to_float(float_to_bf16_bits(f)); }
template <> __device__ __forceinline__ float round_to<f16_t>(float f) { return f16_bits_to_float(float_to_f16_bits(f)); }

// ---------------------------------------------------------------------------------------------------
// K5 (round-3 form).  Workgroup (t, s): `gm_split` workgroups share frame t; every wave ranks the survivors of frame t (one
// ballot per 64 origin slots) and takes the ranks congruent to its id, so the output order is (frame, y1, x1) with no rank pass.
// Built around what the floor probe (tools/micro/floor_probe.hip) showed:
// a two-stage gather of 11.2 k rows of 4 KB runs in 14.6 us when the loads of a stage are all in flight together, while the
// first form spent 22 us: its frame_cnt prefix loop, its bounds branches around the metadata loads and the column geometry of
// a group survivor each became a round trip of its own.  Here
//   * stage 1 is ONE round trip: gcnt / meta / cgeo of the frame's slots with clamped indices (no exec-masked branches), the
//     frame_cnt words of the earlier frames in batches;
//   * a row's chunks (64 bytes per lane) and, for a group survivor, the labels and boxes of the 64 column slots behind it are
//     requested together (the column geometry came with stage 1);
//   * rows go through bounds-checked buffer descriptors (scalar row base, one lane offset, immediates per chunk; lanes past C
//     read zeros and their stores are dropped): no per-lane 64-bit addresses, no exec masks.
// The arithmetic is unchanged: members are added in ascending origin order in the input dtype, one rounding per add.
// ---------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void store_pack_buf(__amdgpu_buffer_rsrc_t rs, uint32_t voff, const Pack<T, VEC>& p) {
    constexpr int bytes = TypeInfo<T>::bytes * VEC;
    if constexpr (bytes == 32) {
        sttm_u32x4 lo, hi;
        __builtin_memcpy(&lo, &p, 16);
        __builtin_memcpy(&hi, reinterpret_cast<const char*>(&p) + 16, 16);
        __builtin_amdgcn_raw_buffer_store_b128(lo, rs, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, rs, voff + 16, 0, 0);
    } else if constexpr (bytes == 16) {
        sttm_u32x4 v;
        __builtin_memcpy(&v, &p, 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
    } else if constexpr (bytes == 8) {
        sttm_u32x2 v;
        __builtin_memcpy(&v, &p, 8);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, 0, 0);
    } else {
        static_assert(bytes == 4, "packs are 4, 8, 16 or 32 bytes");
        unsigned v;
        __builtin_memcpy(&v, &p, 4);
        __builtin_amdgcn_raw_buffer_store_b32(v, rs, voff, 0, 0);
    }
}

struct GmRow { int p, out, n; uint32_t mt, geo; };       // wave-uniform: slot in the frame, output row, group size, box, column geometry

constexpr int kGmLongN = 10;      // COOP: groups of more than this many members are handled by the whole workgroup, after its other rows
constexpr int kGmQueue = 48;      // ... at most this many per workgroup (more: the plain path)

template <typename T, int VEC, int OCC, bool MEM2, bool COOP>
__global__ void __launch_bounds__(256, OCC) k_group_mean(const TemporalArgs a0, const BatchPtrs bp) {
    TemporalArgs a = a0;
    rebase(a, bp, blockIdx.y);
    constexpr int eb = TypeInfo<T>::bytes;
    const int lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ GmRow lq[COOP ? kGmQueue : 1];
    __shared__ int lqn;
    if constexpr (COOP) {
        if (threadIdx.x == 0) lqn = 0;
        __syncthreads();
    }
    const int S = a.gm_split, t = blockIdx.x / S, s = blockIdx.x - t * S;
    const int HW = a.H * a.W;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.bar) {
        const unsigned long long all = __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.bar + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int timed_out = (ld_agent(a.bar + 1) & 2) ? STTM_OVF_BARRIER_TIMEOUT : 0;
        int ovf = (int)(all >> 56);
        if (ovf >= STTM_OVF_BARRIER_TIMEOUT) ovf = STTM_OVF_BARRIER_TIMEOUT - 1;
        publish_counts(a, (int)((all >> 24) & 0xffffffffull), ovf | timed_out);
    }
    const int stride = S * nwave, me = s * nwave + wave;
    const unsigned magicW = a.W > 1 ? 0xffffffffu / (unsigned)a.W + 1u : 0u;         // p / W for p < H * W (p * W < 2^32)
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int NB = 4;                                       // 64-slot chunks whose metadata is loaded together
    constexpr int U0 = 64 / (eb * VEC), U = U0 < 1 ? 1 : (U0 > 8 ? 8 : U0);   // chunks of a row in flight per lane (64 bytes; narrow packs: 8 chunks)
    constexpr int CH = U * 64 * VEC;
    constexpr int NBS = (MEM2 && U <= 4) ? ((COOP || eb == 4) ? 4 : 2) : 1;             // label chunks per step of a survivor's member scan (MEM2 = root cells of > 16 leaves;
                                                                // eight chunks, or four next to eight row pieces per lane, do not fit the registers)
    // rows before this frame: frame_cnt[0 .. t), four words per lane and batch, all requested before the first is used
    int row0 = 0;
    for (int f0 = 0; f0 < t; f0 += 256) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = f0 + k * 64 + lane;
            v[k] = a.frame_cnt[f < t ? f : 0];
            v[k] = f < t ? v[k] : 0;
        }
        row0 += (v[0] + v[1]) + (v[2] + v[3]);
    }
    bool row0_done = false;
    int j0 = 0;
    for (int base = 0; base < HW; base += 64 * NB) {
        int cnt[NB], rank[NB];
        uint32_t meta[NB], geo[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {                          // clamped indices: no branches between the loads
            const int p = base + b * 64 + lane;
            const int pc = p < HW ? p : HW - 1;
            cnt[b] = a.gcnt[t * HW + pc];
            meta[b] = a.meta[t * HW + pc];
            geo[b] = a.cgeo[pc];
        }
        if (!row0_done) { row0 = wave_sum_int(row0); row0_done = true; }
        int nchunk = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (base + b * 64 + lane >= HW) cnt[b] = 0;
            const unsigned long long m = __ballot(cnt[b] > 0);
            rank[b] = j0 + nchunk + __popcll(m & lt);
            nchunk += __popcll(m);
        }
        // the survivor of rank j (j0 <= j < j0 + nchunk): exactly one (chunk, lane) holds it
        auto find = [&](int j) {
            GmRow r;
            int cv = 0; uint32_t mv = 0, gv = 0; int l = 0, bsel = 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const unsigned long long h = __ballot(cnt[b] > 0 && rank[b] == j);
                if (h) { l = __ffsll((long long)h) - 1; bsel = b; }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (bsel == b) { cv = cnt[b]; mv = meta[b]; gv = geo[b]; }
            r.n = __builtin_amdgcn_readlane(cv, l);
            r.mt = (uint32_t)__builtin_amdgcn_readlane((int)mv, l);
            r.geo = (uint32_t)__builtin_amdgcn_readlane((int)gv, l);
            r.p = base + bsel * 64 + l;
            r.out = row0 + j;
            return r;
        };
        int jk = j0 + ((me - j0) & (stride - 1));              // first rank >= j0 congruent to me (stride is a power of two)
        // one survivor: everything from its metadata to its stores.  `acc` arrives loaded with the row's first pass when PRE.
        auto row_desc = [&](const void* basep, int row, int cb) {
            const char* q = reinterpret_cast<const char*>(basep) + ((int64_t)row * a.C + cb) * eb;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(q), 0, (a.C - cb) * eb, 0x00020000);
        };
        auto finish = [&](GmRow r, Pack<T, VEC> (&acc)[U], bool pre) {
#ifdef STTM_DEV
            if (a.dev.k5_mode == 1) r.n = 1;
#endif
            const int n = r.n;
            const int y1 = a.W > 1 ? (int)__umulhi((unsigned)r.p, magicW) : r.p, x1 = r.p - y1 * a.W;
            const int y2 = (int)(r.mt >> 16), x2 = (int)(r.mt & 0xffff);
            const int area = (y2 - y1) * (x2 - x1);
            const int origin = t * HW + r.p;
            int patches = area;
            const void* s0 = (area == 1 && a.xrows) ? a.xrows : a.S;      // 1x1 nodes were not copied out of x
            for (int cb0 = 0; cb0 < a.C; cb0 += CH) {
                // ---- everything this pass needs first: the row's chunks and, for a group survivor, the 64 slots behind it
                if (!(pre && cb0 == 0)) {
                    const __amdgpu_buffer_rsrc_t d = row_desc(s0, origin, cb0);
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[u] = load_pack_buf<T, VEC>(d, (uint32_t)((u * 64 + lane) * VEC * eb));
                }
                if (n > 1) {
                    const Column col = column_from_geo(a, r.geo);
                    const int slot0 = t * col.A + (y1 - col.Y1) * col.aw + (x1 - col.X1);
                    int found = 0;
                    // (NBS label chunks of 64 slots per round trip: on root cells of more than 64 leaves the next frame's members are
                    // A / 64 chunks away -- 16 steps per frame on 6-level trees, where the scan was 26 of the kernel's 58 us in the
                    // development build's ablation, tools/k5_ablate.py modes 1 / 2)
                    for (int sb = slot0 + 1; found < n - 1 && sb < col.slots; sb += 64 * NBS) {
                        int lbs[NBS], mrows[NBS];
                        uint32_t qs[NBS];
#pragma unroll
                        for (int b = 0; b < NBS; ++b) {
                            const int sl = sb + b * 64 + lane;
                            const int slc = sl < col.slots ? sl : col.slots - 1;
                            const int mr0 = slot_to_row(a, col, slc);
                            lbs[b] = a.lab_row[mr0];
                            qs[b] = a.meta[mr0];
                            mrows[b] = sl < col.slots ? mr0 : -1;
                        }
#pragma unroll
                        for (int b = 0; b < NBS; ++b) {
                        const int sl = sb + b * 64 + lane;
                        const int mrow = mrows[b];
                        const uint32_t q = qs[b];
                        const bool hit = mrow >= 0 && lbs[b] == origin;
                        unsigned long long mm = __ballot(hit);
                        if (mm == 0ull) continue;
                        int ar = 0;
                        if (hit) {
                            const int rem = mrow - slot_frame(col, sl) * HW;
                            const int my1 = a.W > 1 ? (int)__umulhi((unsigned)rem, magicW) : rem, mx1 = rem - my1 * a.W;
                            ar = ((int)(q >> 16) - my1) * ((int)(q & 0xffff) - mx1);
                        }
                        // MEM2: members two at a time, both rows requested before the first is added (the sums keep their ascending
                        // order; one member per round trip makes a group of n a chain of n - 1 dependent loads -- heavy merges,
                        // e.g. the 20 x 36 grids at 12 % kept tokens, spend most of this kernel there: 41.6 -> 37.0 us).  It costs
                        // 13 registers, i.e. a wave per SIMD (5 instead of 6): on the 14 x 14 headline, where a group has 1.7 extra
                        // members on average, that is +0.3 us, so the launch picks it for root cells of more than 16 leaves only.
                        while (mm) {
                            const int k0 = __ffsll((long long)mm) - 1;
                            mm &= mm - 1ull;
                            const bool two = MEM2 && mm != 0ull;
                            const int k1 = two ? __ffsll((long long)mm) - 1 : k0;
                            if (two) mm &= mm - 1ull;
                            const int mr0 = __builtin_amdgcn_readlane(mrow, k0), ak0 = __builtin_amdgcn_readlane(ar, k0);
                            const int mr1 = __builtin_amdgcn_readlane(mrow, k1), ak1 = __builtin_amdgcn_readlane(ar, k1);
                            const __amdgpu_buffer_rsrc_t dm0 = row_desc((ak0 == 1 && a.xrows) ? a.xrows : a.S, mr0, cb0);
                            const __amdgpu_buffer_rsrc_t dm1 = row_desc((ak1 == 1 && a.xrows) ? a.xrows : a.S, mr1, cb0);
                            Pack<T, VEC> qv[U], qw[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) qv[u] = load_pack_buf<T, VEC>(dm0, (uint32_t)((u * 64 + lane) * VEC * eb));
                            if (two) {
#pragma unroll
                                for (int u = 0; u < U; ++u) qw[u] = load_pack_buf<T, VEC>(dm1, (uint32_t)((u * 64 + lane) * VEC * eb));
                            }
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const Pack<T, VEC> prev = acc[u];
                                pack_fill(acc[u], [&](int e) { return prev.get(e) + qv[u].get(e); });
                            }
                            ++found;
                            if (cb0 == 0) patches += ak0;
                            if (two) {
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    const Pack<T, VEC> prev = acc[u];
                                    pack_fill(acc[u], [&](int e) { return prev.get(e) + qw[u].get(e); });
                                }
                                ++found;
                                if (cb0 == 0) patches += ak1;
                            }
                        }
                        }   // chunk b
                    }
                }
                if (a.weighted_avg || n > 1) {
                    const float den = round_to<T>(a.weighted_avg ? (float)patches : (float)n);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const Pack<T, VEC> prev = acc[u];
                        pack_fill(acc[u], [&](int e) { return prev.get(e) / den; });
                    }
                }
                const __amdgpu_buffer_rsrc_t dout = row_desc(a.feat_out, r.out, cb0);
#pragma unroll
                for (int u = 0; u < U; ++u) store_pack_buf<T, VEC>(dout, (uint32_t)((u * 64 + lane) * VEC * eb), acc[u]);
            }
            if (a.npatch_out && lane == 0) {
                if (a.idx_out) a.idx_out[r.out] = origin;
                a.npatch_out[r.out] = patches;
                int32_t* o = a.tlbr_out + (int64_t)r.out * 5;
                o[0] = t; o[1] = y1; o[2] = x1; o[3] = y2; o[4] = x2;
            }
        };
        // (two survivors per wave and step with both rows in flight and a quarter of the waves -- all resident at once, a frame's
        // metadata fetched by fewer waves -- was built in round 4 and measured slower in every configuration: headline 23.4 -> 24.8 us,
        // bf16 C=3584 40.4 -> 44.5 us; one row per wave and as many waves as possible stays)
        for (; jk < j0 + nchunk; jk += stride) {
            const GmRow r = find(jk);
            if constexpr (COOP) {
                if (r.n > kGmLongN) {              // (uniform) a long group: queued for the whole workgroup
                    int pos = 0;
                    if (lane == 0) pos = atomicAdd(&lqn, 1);
                    pos = __builtin_amdgcn_readfirstlane(pos);
                    if (pos < kGmQueue) {
                        if (lane == 0) lq[pos] = r;
                        continue;
                    }
                }
            }
            Pack<T, VEC> acc[U];
            finish(r, acc, false);
        }
        j0 += nchunk;
    }
    if constexpr (COOP) {
        // ---- long groups, all waves of the workgroup together ------------------------------------------------------------------
        // A survivor's member rows are a chain of dependent loads in ascending order (the reference's accumulation order,
        // quadtree_temporal_merger.py:123-145): with one wave per survivor a static region that lives through the clip -- 100+ members on
        // the heavily merged 20 x 36 grids -- is 50+ round trips at the end of the kernel while the device drains.  The sums are
        // independent per channel, so the waves split the CHANNELS of such a group (16 bytes per lane: a quarter of a fp32 C = 1024
        // row per wave), every wave adds the members in the same ascending order, scans the column's labels eight chunks of 64
        // slots per round trip and keeps up to eight member chunks in flight.
        __syncthreads();
        const int nq = lqn < kGmQueue ? lqn : kGmQueue;
        constexpr int VC = 16 / eb;                // channels per lane
        constexpr int NBS = 8, MB = 8;
        const unsigned magicWc = a.W > 1 ? 0xffffffffu / (unsigned)a.W + 1u : 0u;
        for (int qi = 0; qi < nq; ++qi) {
            const GmRow r = lq[qi];
            const int n = r.n;
            const int y1 = a.W > 1 ? (int)__umulhi((unsigned)r.p, magicWc) : r.p, x1 = r.p - y1 * a.W;
            const int y2 = (int)(r.mt >> 16), x2 = (int)(r.mt & 0xffff);
            const int area = (y2 - y1) * (x2 - x1);
            const int origin = t * HW + r.p;
            const Column col = column_from_geo(a, r.geo);
            const int slot0 = t * col.A + (y1 - col.Y1) * col.aw + (x1 - col.X1);
            const void* s0 = (area == 1 && a.xrows) ? a.xrows : a.S;
            int patches = area;
            bool first_pass = true;
            for (int cb0 = wave * 64 * VC; cb0 < a.C || first_pass; cb0 += nwave * 64 * VC) {
                // (a wave without channels still walks the scan once: `patches` is wave 0's, but keeping the loop shape uniform is simpler)
                const bool has_ch = cb0 < a.C;
                if (!has_ch && !(first_pass && wave == 0)) break;
                auto desc = [&](const void* basep, int row) {
                    const int cb = has_ch ? cb0 : 0;
                    const char* q = reinterpret_cast<const char*>(basep) + ((int64_t)row * a.C + cb) * eb;
                    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(q), 0, has_ch ? (a.C - cb) * eb : 0, 0x00020000);
                };
                const uint32_t voff = (uint32_t)(lane * VC * eb);
                Pack<T, VC> acc = load_pack_buf<T, VC>(desc(s0, origin), voff);
                int found = 0, pend = 0;
                int prow[MB], parea[MB];
#pragma unroll
                for (int k = 0; k < MB; ++k) { prow[k] = 0; parea[k] = 0; }
                auto flush = [&]() {
                    Pack<T, VC> qv[MB];
#pragma unroll
                    for (int k = 0; k < MB; ++k)
                        if (k < pend) qv[k] = load_pack_buf<T, VC>(desc((parea[k] == 1 && a.xrows) ? a.xrows : a.S, prow[k]), voff);
#pragma unroll
                    for (int k = 0; k < MB; ++k)
                        if (k < pend) {
                            const Pack<T, VC> prev = acc;
                            pack_fill(acc, [&](int e) { return prev.get(e) + qv[k].get(e); });
                        }
                    pend = 0;
                };
                for (int sb = slot0 + 1; found < n - 1 && sb < col.slots; sb += 64 * NBS) {
                    int lbs[NBS], mrows[NBS];
                    uint32_t qs[NBS];
#pragma unroll
                    for (int b = 0; b < NBS; ++b) {
                        const int sl = sb + b * 64 + lane;
                        const int slc = sl < col.slots ? sl : col.slots - 1;
                        const int mr0 = slot_to_row(a, col, slc);
                        lbs[b] = a.lab_row[mr0];
                        qs[b] = a.meta[mr0];
                        mrows[b] = sl < col.slots ? mr0 : -1;
                    }
#pragma unroll
                    for (int b = 0; b < NBS; ++b) {
                        const int sl = sb + b * 64 + lane;
                        const bool hit = mrows[b] >= 0 && lbs[b] == origin;
                        unsigned long long mm = __ballot(hit);
                        if (mm == 0ull) continue;
                        int ar = 0;
                        if (hit) {
                            const int rem = mrows[b] - slot_frame(col, sl) * HW;
                            const int my1 = a.W > 1 ? (int)__umulhi((unsigned)rem, magicWc) : rem, mx1 = rem - my1 * a.W;
                            ar = ((int)(qs[b] >> 16) - my1) * ((int)(qs[b] & 0xffff) - mx1);
                        }
                        while (mm) {
                            const int k0 = __ffsll((long long)mm) - 1;
                            mm &= mm - 1ull;
                            const int mr = __builtin_amdgcn_readlane(mrows[b], k0), ak = __builtin_amdgcn_readlane(ar, k0);
#pragma unroll
                            for (int k = 0; k < MB; ++k)
                                if (k == pend) { prow[k] = mr; parea[k] = ak; }
                            ++pend; ++found;
                            if (first_pass) patches += ak;
                            if (pend == MB) flush();
                        }
                    }
                }
                flush();
                if (has_ch) {
                    const float den = round_to<T>(a.weighted_avg ? (float)patches : (float)n);
                    const Pack<T, VC> prev = acc;
                    pack_fill(acc, [&](int e) { return prev.get(e) / den; });
                    store_pack_buf<T, VC>(desc(a.feat_out, r.out), voff, acc);
                }
                first_pass = false;
            }
            if (wave == 0 && a.npatch_out && lane == 0) {
                if (a.idx_out) a.idx_out[r.out] = origin;
                a.npatch_out[r.out] = patches;
                int32_t* o = a.tlbr_out + (int64_t)r.out * 5;
                o[0] = t; o[1] = y1; o[2] = x1; o[3] = y2; o[4] = x2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Stand-alone temporal stage (sttm_temporal_merge = cross_frame_node_merging_fast / _slow on a caller's node list,
// quadtree_temporal_merger.py:271-299): the node list [N, C] + [N, 5] is brought into the layout the spatial kernel leaves behind --
// features at their origin rows of S, box / inverse norm / default label / default group size by origin row, a node list per
// (frame, root cell), the geometry table, zeroed counters -- and the pair, label and group-mean kernels run unchanged.
// One workgroup per node copies the row and sums its squares on the way (fp32 per lane, fixed-order tree over the lanes and waves).
// ---------------------------------------------------------------------------------------------------
// true iff [y1, y2) x [x1, x2) is the leaf extent of ONE cell of the pyramid `g` (any level from the root level down): walk up from
// the origin leaf and compare the extent of every ancestor
__device__ __forceinline__ bool box_is_tree_cell(const LevelDims& g, int y1, int x1, int y2, int x2) {
    int i = y1, j = x1;
    for (int l = g.n_level - 1; l >= 0; --l) {
        int lo_i = i, hi_i = i, lo_j = j, hi_j = j;
        for (int m = l; m < g.n_level - 1; ++m) {
            lo_i = child_start(lo_i, g.h[m + 1]);
            hi_i = child_start(hi_i, g.h[m + 1]) + child_count(hi_i, g.h[m + 1]) - 1;
            lo_j = child_start(lo_j, g.w[m + 1]);
            hi_j = child_start(hi_j, g.w[m + 1]) + child_count(hi_j, g.w[m + 1]) - 1;
        }
        if (lo_i == y1 && lo_j == x1 && hi_i + 1 == y2 && hi_j + 1 == x2) return true;
        if (lo_i != y1 || lo_j != x1) return false;          // a coarser ancestor no longer starts at this leaf
        if (l > 0) { i = parent_of(i, g.h[l]); j = parent_of(j, g.w[l]); }
    }
    return false;
}

// One workgroup per node of the caller's list.  A node is taken only if (a) its box lies inside the grid, (b) the box is a cell of the
// root_level partition (what quadtree_build_video emits), (c) none of its leaves is covered by another node of the list -- every node
// marks its leaves in `cover` with atomic adds, so duplicated origins and overlapping boxes are seen by at least one of the two nodes --
// and (d) its root cell's list has room.  A rejected node writes nothing but the sticky overflow flag (bar[1] -> counts[STTM_CNT_OVERFLOW],
// "outputs invalid"); the tables the later kernels walk stay consistent (list heads count stored entries only, one node per origin,
// disjoint boxes), so those kernels stay inside their buffers whatever the caller passed.
template <typename T>
__global__ void __launch_bounds__(256) k_ingest_nodes(const TemporalArgs a, const void* feat, const int32_t* tlbr, int n_nodes,
                                                      void* S, uint32_t* meta, double* inrm, int* rc_list, int32_t* cover) {
    __shared__ float wsum[4];
    __shared__ int bad;
    const int i = blockIdx.x;
    if (i >= n_nodes) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = tlbr[5 * i], y1 = tlbr[5 * i + 1], x1 = tlbr[5 * i + 2], y2 = tlbr[5 * i + 3], x2 = tlbr[5 * i + 4];
    const bool ok = t >= 0 && t < a.T && y1 >= 0 && x1 >= 0 && y2 > y1 && x2 > x1 && y2 <= a.H && x2 <= a.W &&
                    box_is_tree_cell(a.dims, y1, x1, y2, x2);
    if (!ok) {                          // (uniform) not a cell of the partition: the call reports an overflow, nothing is written
        if (tid == 0) __hip_atomic_fetch_or(a.bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    const int bw = x2 - x1, area = (y2 - y1) * bw;
    for (int q = tid; q < area; q += 256) {
        const int ly = q / bw, lx = q - ly * bw;
        if (atomicAdd(cover + (int64_t)t * a.H * a.W + (y1 + ly) * a.W + (x1 + lx), 1) != 0) bad = 1;
    }
    __syncthreads();
    if (bad) {                          // (uniform) a duplicated origin or an overlapping box
        if (tid == 0) __hip_atomic_fetch_or(a.bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int64_t row = (int64_t)t * a.H * a.W + y1 * a.W + x1;
    const int rcell = root_cell_of(a.dims, y1, x1);
    int* list = rc_list + ((int64_t)t * a.R + rcell) * a.rc_stride;
    const bool one = y2 - y1 == 1 && x2 - x1 == 1;
    if (tid == 0) {
        // the list head counts STORED entries only: an entry that finds the list full takes its increment back (no node is
        // ever stored at or beyond rc_stride - 1, and the later kernels never read a count above it)
        const int inc = 1 + (one ? 1 << 16 : 0);
        const int pos = atomicAdd(list, inc) & 0xffff;
        if (pos < a.rc_stride - 1) list[1 + pos] = (y1 << 24) | (x1 << 16) | (y2 << 8) | x2;
        else { atomicSub(list, inc); bad = 1; }
    }
    __syncthreads();
    if (bad) {                          // more nodes than leaves in a root cell
        if (tid == 0) __hip_atomic_fetch_or(a.bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    float acc = 0.f;
    for (int c = tid; c < a.C; c += 256) {
        float v;
        if constexpr (TypeInfo<T>::bytes == 4) {
            v = reinterpret_cast<const float*>(feat)[(int64_t)i * a.C + c];
            reinterpret_cast<float*>(S)[row * a.C + c] = v;
        } else {
            const uint16_t bits = reinterpret_cast<const uint16_t*>(feat)[(int64_t)i * a.C + c];
            reinterpret_cast<uint16_t*>(S)[row * a.C + c] = bits;
            if constexpr (std::is_same<T, bf16_t>::value) v = bf16_bits_to_float(bits);
            else v = f16_bits_to_float(bits);
        }
        acc += v * v;
    }
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        const float n2 = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        meta[row] = ((uint32_t)y2 << 16) | (uint32_t)x2;
        inrm[row] = 1.0 / (sqrt((double)n2) + 1e-8);
        a.lab_row[row] = (int32_t)row;
        a.gcnt[row] = 1;
    }
}
// geometry table + the counters the spatial kernel zeroes
__global__ void __launch_bounds__(256) k_ingest_geo(const TemporalArgs a, uint32_t* cgeo) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < a.H * a.W) {
        const int y = p / a.W, x = p - y * a.W;
        const int r = root_cell_of(a.dims, y, x);
        const Column c = make_column(a, a.dims, r);
        cgeo[p] = (uint32_t)c.Y1 | ((uint32_t)c.X1 << 8) | ((uint32_t)c.aw << 16) | ((uint32_t)c.ah << 24);
    }
    if (p < a.T) a.frame_cnt[p] = 0;
    if (p < a.R) a.col_arrive[p] = 0;
    if (p < STTM_CNT_SLOTS) a.counts[p] = 0;
}
hipError_t launch_ingest_nodes(const TemporalArgs& a, const void* feat, const int32_t* tlbr, int n_nodes, void* S, uint32_t* meta,
                               double* inrm, int* rc_list, uint32_t* cgeo, hipStream_t stream) {
    const size_t N = (size_t)a.T * a.H * a.W;
    hipError_t e;
    // every origin row starts as "no node here"; the node lists start empty; bar: overflow flag / N' word / barrier word
    if ((e = hipMemsetAsync(meta, 0, N * 4, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(a.gcnt, 0, N * 4, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(a.lab_row, 0xff, N * 4, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(rc_list, 0, (size_t)a.T * a.R * a.rc_stride * 4, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(a.bar, 0, 32, stream)) != hipSuccess) return e;
    // leaf cover marks of the ingest (the label scratch is >= 10 N ints and is written by the label stage before it reads it)
    int32_t* cover = a.colscratch;
    if ((e = hipMemsetAsync(cover, 0, N * 4, stream)) != hipSuccess) return e;
    int most = a.H * a.W;
    if (a.T > most) most = a.T;
    if (a.R > most) most = a.R;
    hipLaunchKernelGGL(k_ingest_geo, dim3((most + 255) / 256), dim3(256), 0, stream, a, cgeo);
    if (n_nodes > 0) {
        if (a.dtype == STTM_F32) hipLaunchKernelGGL((k_ingest_nodes<float>), dim3(n_nodes), dim3(256), 0, stream, a, feat, tlbr, n_nodes, S, meta, inrm, rc_list, cover);
        else if (a.dtype == STTM_BF16) hipLaunchKernelGGL((k_ingest_nodes<bf16_t>), dim3(n_nodes), dim3(256), 0, stream, a, feat, tlbr, n_nodes, S, meta, inrm, rc_list, cover);
        else hipLaunchKernelGGL((k_ingest_nodes<f16_t>), dim3(n_nodes), dim3(256), 0, stream, a, feat, tlbr, n_nodes, S, meta, inrm, rc_list, cover);
    }
    return hipGetLastError();
}

hipError_t launch_group_mean(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream) {
    const int grid = a.T * a.gm_split;
    const bool mem2 = a.max_slots > 16 * a.T;          // root cells of more than 16 leaves (4-level trees and deeper)
    // root cells of more than 64 leaves (5- / 6-level trees): long groups go to the whole workgroup (COOP).  Same-box A/B
    // (profiles/r04z_k5_coop_ab.txt): 27 x 27 root 0, T = 64: K5 88.5 -> 64.0 us; 36 x 64 root 0, T = 16: 40.5 -> 41.3; on the
    // 64-leaf root cells of the 20 x 36 / 18 x 26 grids 37.0 -> 39.4 / 40.1 -> 39.6 us -- their member time is many medium groups,
    // not a few long ones --, so those keep the plain two-members-in-flight kernel.
    const bool deep = a.max_slots > 64 * a.T;
#define STTM_LAUNCH_GM(TT, VV) do { if (deep) hipLaunchKernelGGL((k_group_mean<TT, VV, 4, true, true>), dim3(grid, n_videos), dim3(256), 0, stream, a, bp); \
                                    else if (mem2) hipLaunchKernelGGL((k_group_mean<TT, VV, 5, true, false>), dim3(grid, n_videos), dim3(256), 0, stream, a, bp); \
                                    else hipLaunchKernelGGL((k_group_mean<TT, VV, 6, false, false>), dim3(grid, n_videos), dim3(256), 0, stream, a, bp); } while (0)
    if (a.dtype == STTM_F32) {
        if (a.vec == 8) STTM_LAUNCH_GM(float, 8); else if (a.vec == 4) STTM_LAUNCH_GM(float, 4); else if (a.vec == 2) STTM_LAUNCH_GM(float, 2); else STTM_LAUNCH_GM(float, 1);
    } else if (a.dtype == STTM_BF16) {
        if (a.vec == 8) STTM_LAUNCH_GM(bf16_t, 8); else if (a.vec == 4) STTM_LAUNCH_GM(bf16_t, 4); else STTM_LAUNCH_GM(bf16_t, 2);
    } else {
        if (a.vec == 8) STTM_LAUNCH_GM(f16_t, 8); else if (a.vec == 4) STTM_LAUNCH_GM(f16_t, 4); else STTM_LAUNCH_GM(f16_t, 2);
    }
#undef STTM_LAUNCH_GM
    return hipGetLastError();
}

}  // namespace sttm
