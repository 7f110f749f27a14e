// K2..K5 -- temporal stage of STTM on gfx950.
//
//   k_pairs        candidate pairs + cosine filter            (quadtree_temporal_merger.py:8-73 of the reference)
//   k_slow_filter  slow_ver: per frame pair, similarity sort + adjacent-duplicate removal                (:75-121)
//   k_col_labels   label propagation per root-cell column, in LDS                                      (:223-269)
//                  <FUSED>: probe -> grid barrier -> exact replay -> groups -> arrival word (N' to the host), one launch
//                  <PROBE> / <FINAL>: the same code as two launches when co-residency is not guaranteed
//   k_group_mean   ranks the survivors of its frame, then per-survivor ascending-order accumulation and mean (:123-171)
//
// All of them address nodes by their ORIGIN ROW  t*H*W + y1*W + x1  in the scratch matrix S written by the
// spatial kernel (1x1 nodes stay in x).  Origin rows are ordered exactly like the reference's sorted node
// indices, so min-label propagation over origin rows is the same computation as over node indices.
//
// Structure that makes this cheap: a node lies inside exactly one root cell and root cells are the same in
// every frame, so (a) candidate pairs never cross root cells -- one workgroup per (frame pair, root cell)
// enumerates <= 16x16 box tests instead of the reference's dense [T-1, M, M, 4] tensor -- and (b) the label
// graph splits into R independent columns (one per root cell, T frames deep) that fit in LDS.  The reference's
// loop is synchronous and stops at the first iteration where ALL labels are idempotent (quirk Q2: that is not
// connected components), so the columns must all run the same number of iterations: every column probes to its
// fixed point and records after which iterations it was idempotent; the global count K is the first iteration
// at which all columns were; a column whose fixed point came later replays exactly K iterations.
#include <cstdlib>

#include "sttm_kernels.h"

namespace sttm {

__device__ __forceinline__ int ld_agent(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int32_t* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// Columns (used by the pair kernel to emit local slot ids and by the label kernels).
// A column = root cell (I, J) over all T frames.  Local slot of leaf (t, y, x) inside the column:
//   s = t*A + (y - Y1)*aw + (x - X1),   A = area of the root cell in leaves; monotone in the origin row.
// ---------------------------------------------------------------------------------------------------
struct Column {
    int Y1, X1, ah, aw, A, slots, base;   // base = T * (leaves of the root cells before this one)
    unsigned mA, maw;                     // ceil(2^32 / A), ceil(2^32 / aw): exact n / d for n * d < 2^32 (slots <= 65536)
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
__device__ __forceinline__ Column make_column(const TemporalArgs& a, int r) {
    const LevelDims& g = a.dims;
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
    return c;
}
__device__ __forceinline__ int slot_to_row(const TemporalArgs& a, const Column& c, int s) {
    const int t = c.A == 1 ? s : (int)__umulhi((unsigned)s, c.mA), q = s - t * c.A;
    const int ly = c.aw == 1 ? q : (int)__umulhi((unsigned)q, c.maw), lx = q - ly * c.aw;
    return t * a.H * a.W + (c.Y1 + ly) * a.W + (c.X1 + lx);
}
__device__ __forceinline__ int row_to_slot(const TemporalArgs& a, const Column& c, int row) {
    const int HW = a.H * a.W;
    const int t = row / HW, rem = row - t * HW;
    const int y = rem / a.W, x = rem - y * a.W;
    return t * c.A + (y - c.Y1) * c.aw + (x - c.X1);
}

// ---------------------------------------------------------------------------------------------------
// K2: pairs.  One workgroup per (t, root cell): box tests between the node lists of frames t and t+1,
// then one wave per candidate for the C-long dot product (two candidates in flight per wave).
// Kept edges go to this workgroup's own slot list -- no global counters.
// ---------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_pairs(TemporalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* cand = reinterpret_cast<int*>(smem_raw);     // [cap] packed (ia << 16 | ib)
    __shared__ int ncand, nkept;
#define STTM_K2_TICK(n) do { if (a.dbg_ticks_k2 && blockIdx.x == a.dbg_wg_k2 && threadIdx.x == 0) a.dbg_ticks_k2[n] = wall_clock64(); } while (0)
    STTM_K2_TICK(0);
    if (a.dbg_ticks_k2 && blockIdx.x == 0 && threadIdx.x == 0) a.dbg_ticks_k2[8] = wall_clock64();
    const int R = a.R;
    int t, r;
    if (a.pairs_seg > 0) {
        // XCD-aware map.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2, and the node rows of
        // (t+1, r) are read by the workgroups of pair t and of pair t+1: keep a run of consecutive frames of one root
        // cell on ONE XCD so that the second read hits that L2 instead of going to the fabric again.
        const int L = a.pairs_seg, nf = a.T - 1;
        const int segs = (nf + L - 1) / L;
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int chunk = (i / L) * 8 + xcd, pos = i % L;
        if (chunk >= R * segs) return;
        r = chunk / segs;
        t = (chunk - r * segs) * L + pos;
        if (t >= nf) return;
    } else {
        t = blockIdx.x / R; r = blockIdx.x % R;
    }
    const int HW = a.H * a.W;
    const int* LA = a.rc_list + (int64_t)(t * R + r) * a.rc_stride;
    const int* LB = a.rc_list + (int64_t)((t + 1) * R + r) * a.rc_stride;
    const int cap = a.ecap;
    const Column col = make_column(a, r);
    const int64_t cidx = (int64_t)r * (a.T - 1) + t;          // column-major: a column's lists are contiguous
    int32_t* my_edges = a.edges + cidx * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    // ONE round trip for the counts and the node boxes of both cells: every list entry is fetched speculatively (entries past
    // the count are stale but inside the row) and parked in LDS for the box tests
    int* listA = cand + cap;                              // [rc_stride]
    int* listB = listA + a.rc_stride;                     // [rc_stride]
    for (int i = tid; i < a.rc_stride; i += blockDim.x) { listA[i] = LA[i]; listB[i] = LB[i]; }
    if (tid == 0) { ncand = 0; nkept = 0; }
    __syncthreads();
    const int nA = listA[0], nB = listB[0];
    // box tests on a (ia, ib) grid whose width is the power of two >= nB: no integer division per test
    int lg = 0;
    while ((1 << lg) < nB) ++lg;
    const int ib = tid & ((1 << lg) - 1), ia0 = tid >> lg, ia_step = blockDim.x >> lg;
    if (ib < nB && ia_step > 0) {
        const unsigned bb = (unsigned)listB[1 + ib];
        const int by1 = bb >> 24, bx1 = (bb >> 16) & 255, by2 = (bb >> 8) & 255, bx2 = bb & 255;
        for (int ia = ia0; ia < nA; ia += ia_step) {
            const unsigned ba = (unsigned)listA[1 + ia];
            const int ay1 = ba >> 24, ax1 = (ba >> 16) & 255, ay2 = (ba >> 8) & 255, ax2 = ba & 255;
            const bool a_has_b = ay1 <= by1 && ax1 <= bx1 && ay2 >= by2 && ax2 >= bx2;
            const bool b_has_a = ay1 >= by1 && ax1 >= bx1 && ay2 <= by2 && ax2 <= bx2;
            if (a_has_b || b_has_a) {
                const int pos = atomicAdd(&ncand, 1);
                if (pos < cap) cand[pos] = (ia << 16) | ib;
            }
        }
    } else if (ia_step == 0) {                            // more nodes in the cell than threads (block-size override): plain loop
        for (int p = tid; p < nA * nB; p += blockDim.x) {
            const int ia = p / nB, jb = p - ia * nB;
            const unsigned ba = (unsigned)listA[1 + ia], bb = (unsigned)listB[1 + jb];
            const int ay1 = ba >> 24, ax1 = (ba >> 16) & 255, ay2 = (ba >> 8) & 255, ax2 = ba & 255;
            const int by1 = bb >> 24, bx1 = (bb >> 16) & 255, by2 = (bb >> 8) & 255, bx2 = bb & 255;
            if ((ay1 <= by1 && ax1 <= bx1 && ay2 >= by2 && ax2 >= bx2) || (ay1 >= by1 && ax1 >= bx1 && ay2 <= by2 && ax2 <= bx2)) {
                const int pos = atomicAdd(&ncand, 1);
                if (pos < cap) cand[pos] = (ia << 16) | jb;
            }
        }
    }
    __syncthreads();
    STTM_K2_TICK(1);
    const int nc = ncand < cap ? ncand : cap;
    auto row_of = [&](const int* L, int i, int frame) {
        const unsigned b = (unsigned)L[1 + i];
        return frame * HW + (int)(b >> 24) * a.W + (int)((b >> 16) & 255);
    };
    // column-local slot of a list entry, straight from its box (no division: row_to_slot would divide by H*W and W)
    auto slot_of = [&](const int* L, int i, int frame) {
        const unsigned b = (unsigned)L[1 + i];
        return frame * col.A + ((int)(b >> 24) - col.Y1) * col.aw + ((int)((b >> 16) & 255) - col.X1);
    };
    auto src_of = [&](const int* L, int i) -> const void* {      // 1x1 nodes were not copied out of x
        const unsigned b = (unsigned)L[1 + i];
        const bool leaf = ((b >> 8) & 255) - (b >> 24) == 1 && (b & 255) - ((b >> 16) & 255) == 1;
        return (leaf && a.xrows) ? a.xrows : a.S;
    };
    if (a.n_head > 0) {
        // per-head cosine, averaged over heads (quadtree_temporal_merger.py:65-68): G adjacent lanes own one head
        const int G = a.head_lanes;
        for (int c = wave; c < nc; c += nwave) {
            const int k0 = cand[c];
            const int rowA = row_of(listA, k0 >> 16, t), rowB = row_of(listB, k0 & 0xffff, t + 1);
            const void* sA = src_of(listA, k0 >> 16); const void* sB = src_of(listB, k0 & 0xffff);
            float acc = 0.f;
            for (int base = 0; base < a.C; base += 64 * VEC) {
                const int c0 = base + lane * VEC;
                float d = 0.f, na = 0.f, nb = 0.f;
                if (c0 < a.C) {
                    const Pack<T, VEC> pa = load_pack<T, VEC>(sA, (int64_t)rowA * a.C + c0);
                    const Pack<T, VEC> pb = load_pack<T, VEC>(sB, (int64_t)rowB * a.C + c0);
                    d = dot_pack(pa, pb); na = dot_pack(pa, pa); nb = dot_pack(pb, pb);
                }
                for (int m = 1; m < G; m <<= 1) {
                    d += __shfl_xor(d, m, 64); na += __shfl_xor(na, m, 64); nb += __shfl_xor(nb, m, 64);
                }
                if ((lane & (G - 1)) == 0 && c0 < a.C) acc += d / ((sqrtf(na) + 1e-8f) * (sqrtf(nb) + 1e-8f));
            }
            acc = wave_sum(acc);
            if (lane == 0 && acc / (float)a.n_head >= a.temporal_thresh) {
                const int e = atomicAdd(&nkept, 1);
                my_edges[e] = (int)(((unsigned)slot_of(listA, k0 >> 16, t) << 16) | (unsigned)slot_of(listB, k0 & 0xffff, t + 1));
            }
        }
    } else
    for (int c = wave; c < nc; c += 2 * nwave) {
        const int c2 = c + nwave;
        const bool two = c2 < nc;
        const int k0 = cand[c], k1 = two ? cand[c2] : k0;
        const int rowA0 = row_of(listA, k0 >> 16, t), rowB0 = row_of(listB, k0 & 0xffff, t + 1);
        const int rowA1 = row_of(listA, k1 >> 16, t), rowB1 = row_of(listB, k1 & 0xffff, t + 1);
        const void* sA0 = src_of(listA, k0 >> 16); const void* sB0 = src_of(listB, k0 & 0xffff);
        const void* sA1 = src_of(listA, k1 >> 16); const void* sB1 = src_of(listB, k1 & 0xffff);
        // the two lanes that finish the cosines fetch their inverse norms now, under the row loads
        double pre_ia = 0.0, pre_ib = 0.0;
        if (lane < 2 && !a.inline_norms) {
            pre_ia = a.inrm[lane ? rowA1 : rowA0];
            pre_ib = a.inrm[lane ? rowB1 : rowB0];
        }
        float d0 = 0.f, d1 = 0.f, na0 = 0.f, nb0 = 0.f, na1 = 0.f, nb1 = 0.f;
        for (int c0 = lane * VEC; c0 < a.C; c0 += 64 * VEC) {
            const Pack<T, VEC> pa0 = load_pack<T, VEC>(sA0, (int64_t)rowA0 * a.C + c0);
            const Pack<T, VEC> pb0 = load_pack<T, VEC>(sB0, (int64_t)rowB0 * a.C + c0);
            const Pack<T, VEC> pa1 = load_pack<T, VEC>(sA1, (int64_t)rowA1 * a.C + c0);
            const Pack<T, VEC> pb1 = load_pack<T, VEC>(sB1, (int64_t)rowB1 * a.C + c0);
            d0 += dot_pack(pa0, pb0);
            d1 += dot_pack(pa1, pb1);
            if (a.inline_norms) {           // the per-head spatial kernel does not produce whole-vector norms
                na0 += dot_pack(pa0, pa0); nb0 += dot_pack(pb0, pb0);
                na1 += dot_pack(pa1, pa1); nb1 += dot_pack(pb1, pb1);
            }
        }
        d0 = wave_sum(d0);
        d1 = wave_sum(d1);
        if (a.inline_norms) { na0 = wave_sum(na0); nb0 = wave_sum(nb0); na1 = wave_sum(na1); nb1 = wave_sum(nb1); }
        if (lane < 2 && (lane == 0 || two)) {
            const float dot = lane ? d1 : d0;
            // x / (|x| + 1e-8) on both sides (quadtree_temporal_merger.py:62-63); the spatial kernel stored
            // 1 / (|x| + 1e-8) in double
            const double ia = a.inline_norms ? 1.0 / (sqrt((double)(lane ? na1 : na0)) + 1e-8) : pre_ia;
            const double ib = a.inline_norms ? 1.0 / (sqrt((double)(lane ? nb1 : nb0)) + 1e-8) : pre_ib;
            const float sim = (float)((double)dot * ia * ib);
            if (sim >= a.temporal_thresh) {
                const int e = atomicAdd(&nkept, 1);
                if (a.edge_sim) a.edge_sim[cidx * cap + e] = sim;
                const int kk = lane ? k1 : k0;
                my_edges[e] = (int)(((unsigned)slot_of(listA, kk >> 16, t) << 16) | (unsigned)slot_of(listB, kk & 0xffff, t + 1));
            }
        }
    }
    __syncthreads();
    STTM_K2_TICK(2);
    for (int e = nkept + tid; e < cap; e += blockDim.x) my_edges[e] = -1;      // unused entries: the reader scans all `cap`
    if (tid == 0) {
        a.edge_cnt[cidx] = nkept;
        a.cand_cnt[cidx] = ncand;
    }
    STTM_K2_TICK(3);
    if (a.dbg_ticks_k2 && blockIdx.x == a.dbg_wg_k2 && threadIdx.x == 0) a.dbg_ticks_k2[4] = ncand;
#undef STTM_K2_TICK
}

hipError_t launch_pairs(const TemporalArgs& a, hipStream_t stream) {
    if (a.T < 2) return hipSuccess;
    int grid = (a.T - 1) * a.R;
    if (a.pairs_seg > 0) {
        const int segs = (a.T - 1 + a.pairs_seg - 1) / a.pairs_seg;
        grid = 8 * ((a.R * segs + 7) / 8) * a.pairs_seg;
    }
    const size_t smem = sizeof(int) * ((size_t)a.ecap + 2 * (size_t)a.rc_stride);      // candidates + the two node lists
    static const int nt_env = [] { const char* e = getenv("STTM_PAIRS_NT"); const int v = e ? atoi(e) : 0; return (v == 64 || v == 128 || v == 256) ? v : 0; }();
    const int nt = nt_env ? nt_env : 256;
#define STTM_LAUNCH_PAIRS(TT, VV) hipLaunchKernelGGL((k_pairs<TT, VV>), dim3(grid), dim3(nt), smem, stream, a)
    if (a.dtype == STTM_F32) {
        if (a.vec == 8) STTM_LAUNCH_PAIRS(float, 8); else if (a.vec == 4) STTM_LAUNCH_PAIRS(float, 4); else if (a.vec == 2) STTM_LAUNCH_PAIRS(float, 2); else STTM_LAUNCH_PAIRS(float, 1);
    } else if (a.dtype == STTM_BF16) {
        if (a.vec == 8) STTM_LAUNCH_PAIRS(bf16_t, 8); else if (a.vec == 4) STTM_LAUNCH_PAIRS(bf16_t, 4); else STTM_LAUNCH_PAIRS(bf16_t, 2);
    } else {
        if (a.vec == 8) STTM_LAUNCH_PAIRS(f16_t, 8); else if (a.vec == 4) STTM_LAUNCH_PAIRS(f16_t, 4); else STTM_LAUNCH_PAIRS(f16_t, 2);
    }
#undef STTM_LAUNCH_PAIRS
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------
// slow_ver (get_cross_frame_node_pairs_slow, quadtree_temporal_merger.py:75-121): per frame pair, the kept edges of
// ALL root cells are ordered by similarity, descending, and an edge is dropped when its src (later-frame node)
// equals the src of the edge right before it in that order -- adjacent-duplicate removal, not an arg-max per src.
// One workgroup per frame pair: gather -> bitonic sort in LDS -> filter -> rewrite the per-column lists.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_slow_filter(TemporalArgs a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* key = reinterpret_cast<float*>(smem_raw);            // [npad] similarity
    int* ent = reinterpret_cast<int*>(key + npad);               // [npad] packed local slots (dst << 16 | src)
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
    // drop an edge whose src equals the previous edge's src (same root cell and same src slot)
    for (int i = tid; i < n; i += nt) {
        const bool dup = i > 0 && colr[i] == colr[i - 1] && (ent[i] & 0xffff) == (ent[i - 1] & 0xffff);
        if (!dup) {
            const int r = colr[i];
            const int pos = atomicAdd(&ccnt[r], 1);
            a.edges[((int64_t)r * nf + t) * cap + pos] = ent[i];
        }
    }
    __syncthreads();
    for (int j = tid; j < R * cap; j += nt) {
        const int r = j / cap, e = j - r * cap;
        if (e >= ccnt[r]) a.edges[((int64_t)r * nf + t) * cap + e] = -1;
    }
    for (int r = tid; r < R; r += nt) a.edge_cnt[(int64_t)r * nf + t] = ccnt[r];
}

hipError_t launch_slow_filter(const TemporalArgs& a, hipStream_t stream) {
    if (a.T < 2) return hipSuccess;
    int npad = 2;
    while (npad < a.R * a.ecap) npad <<= 1;                       // worst case: every list full
    const size_t smem = sizeof(int) * ((size_t)3 * npad + a.R);
    if (smem > 150 * 1024) return hipErrorInvalidValue;
    int nthreads = npad < 1024 ? (npad < 64 ? 64 : npad) : 1024;
    hipLaunchKernelGGL(k_slow_filter, dim3(a.T - 1), dim3(nthreads), smem, stream, a, npad);
    return hipGetLastError();
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

__device__ __forceinline__ int box_area(uint32_t meta, int row, int HW, int W) {
    const int y2 = meta >> 16, x2 = meta & 0xffff;
    const int rem = row % HW;
    const int y1 = rem / W, x1 = rem - y1 * W;
    return (y2 - y1) * (x2 - x1);
}

constexpr int kColThreads = 1024;
constexpr int kLeafBit = (int)0x80000000;     // members: the row is a 1x1 node (its feature lives in x, not S)
constexpr int kMaxProbeIters = 62;

// Column arrays live in LDS (GMEM == false) or, for columns too large for the 160 KB LDS, in a per-column
// slice of global scratch (GMEM == true).  In global memory every access is agent scope (L2-served), so
// phases separated by __syncthreads() see each other's plain writes and atomics alike.
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
// node areas: uint16 in LDS, int32 (agent scope) in the global-memory fallback
template <bool GMEM> struct AreaT { typedef uint16_t type; };
template <> struct AreaT<true> { typedef int type; };
template <bool GMEM> __device__ __forceinline__ int carea_ld(const typename AreaT<GMEM>::type* p, int i) {
    if constexpr (GMEM) return ld_agent(p + i); else return p[i];
}
template <bool GMEM> __device__ __forceinline__ void carea_st(typename AreaT<GMEM>::type* p, int i, int v) {
    if constexpr (GMEM) st_agent(p + i, v); else p[i] = (uint16_t)v;
}
template <bool GMEM> __device__ __forceinline__ int caadd(int* p, int v) {
    if constexpr (GMEM) return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return atomicAdd(p, v);
}

// One synchronous iteration of get_merge_dst_idx_safe.
//   rep  : labels before (read only during the edge pass)       rep2 : copy of rep that receives the amin scatter
// after the call rep == rep2 == new labels.  flags[0] = some label changed, flags[1] = not idempotent.
// Barrier between the phases of a column workgroup.  LDS-resident columns exchange everything through LDS, so their
// barriers must not wait for the wave's outstanding global stores (group tables, member lists: consumed by the NEXT
// kernel) -- __syncthreads() would drain them every time.  Columns spilled to global scratch need the full barrier.
template <bool GMEM> __device__ __forceinline__ void col_sync() {
    if constexpr (GMEM) __syncthreads(); else lds_barrier();
}

template <bool GMEM>
__device__ __forceinline__ void column_iteration(int* rep, int* rep2, const int* edges, int E, int slots, int* flags) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { flags[0] = 0; flags[1] = 0; }
    for (int e = tid; e < E; e += nt) {
        const unsigned pr = (unsigned)cld<GMEM>(edges + e);
        const int d = pr >> 16, s = pr & 0xffffu;
        const int rd = cld<GMEM>(rep + d), rs = cld<GMEM>(rep + s);
        const int m = rd < rs ? rd : rs;
        camin<GMEM>(rep2 + d, m);
        camin<GMEM>(rep2 + s, m);
    }
    col_sync<GMEM>();
    int changed = 0;
    for (int i = tid; i < slots; i += nt) {
        const int v = cld<GMEM>(rep2 + cld<GMEM>(rep2 + i));
        if (v != cld<GMEM>(rep + i)) changed = 1;
        cst<GMEM>(rep + i, v);
    }
    if (changed) flags[0] = 1;
    col_sync<GMEM>();
    int bad = 0;
    for (int i = tid; i < slots; i += nt) {
        const int v = cld<GMEM>(rep + i);
        cst<GMEM>(rep2 + i, v);
        if (cld<GMEM>(rep + v) != v) bad = 1;
    }
    if (bad) flags[1] = 1;
    col_sync<GMEM>();
}

// Grid-wide barrier among the (co-resident) column workgroups of the fused kernel.  The spin is bounded: on a
// timeout the overflow counter is raised (the Python wrapper then fails loudly) instead of hanging the GPU.
__device__ __forceinline__ void grid_barrier(int32_t* counter, int target, int32_t* overflow) {
    // Everything that crosses this barrier is written with agent-scope (write-through, sc1) stores or atomics and read
    // with agent-scope loads, so no release/acquire cache maintenance is needed: every wave drains its stores, one lane
    // arrives and polls the counter.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();                       // 100 MHz
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000ll) { atomicAdd(overflow, 1); break; }    // 0.2 s
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void publish_counts(const TemporalArgs& a, int n_out) {
    a.counts[STTM_CNT_OUT] = n_out;
    if (a.counts_host) {
        // straight into pinned host memory: the caller learns N' while k_group_mean still runs
        for (int i = 0; i < STTM_CNT_SLOTS - 1; ++i)
            __hip_atomic_store(a.counts_host + i, i == STTM_CNT_OUT ? n_out : ld_agent(a.counts + i), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.counts_host + STTM_CNT_SLOTS - 1, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

enum { COL_PROBE = 0, COL_FINAL = 1, COL_FUSED = 2 };

#define STTM_TICK(n) do { if (a.dbg_ticks && blockIdx.x == a.dbg_wg && threadIdx.x == 0) a.dbg_ticks[n] = wall_clock64(); } while (0)


template <int MODE, bool GMEM>
__global__ void __launch_bounds__(kColThreads) k_col_labels(TemporalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int flags[4];         // two (changed, not-idempotent) pairs, used by alternate iterations
    __shared__ int wsum[16];
    __shared__ unsigned long long wmask[16];
    const int r = blockIdx.x, R = a.R;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
    STTM_TICK(0);
    if (a.dbg_ticks && blockIdx.x == a.dbg_wg && threadIdx.x == 0) a.dbg_ticks[12] = clock64();
    const Column col = make_column(a, r);
    const int slots = col.slots;
    // four arrays of `slots` ints each
    int* A0 = GMEM ? a.colscratch + (int64_t)5 * col.base : reinterpret_cast<int*>(smem_raw);
    int* rep = A0;                 // labels
    int* rep2 = A0 + slots;        // scatter target during iterations; group sizes afterwards
    int* edges = A0 + 2 * slots;   // [E] packed local (dst << 16 | src) during iterations (E <= 2 * slots)
    int* aux = A0 + 2 * slots;     // afterwards: group offsets / fill cursor
    int* mem = A0 + 3 * slots;     // afterwards: unordered member lists
    typename AreaT<GMEM>::type* area_l = reinterpret_cast<typename AreaT<GMEM>::type*>(A0 + 4 * slots);   // [slots]
    const bool temporal = a.temporal_thresh > 0.f && a.T > 1;

    // ---- gather this column's edges (local slot ids) ------------------------------------------------------
    // Every global load of this phase (first round of the edge lists, node boxes, candidate counts) is issued before the
    // first workgroup barrier and consumed after it: one memory round trip for the whole phase.
    __shared__ int ecount;
    int E = 0;
    constexpr int PER = 8;
    const int nf = a.T - 1;
    const int32_t* elist = a.edges + (int64_t)r * nf * a.ecap;
    const int total = temporal ? nf * a.ecap : 0;
    int local[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {                         // independent, coalesced loads
        const int j = k * nt + tid;
        local[k] = j < total ? elist[j] : -1;
    }
    uint32_t my_meta[4];
    int my_row[4];
    const bool few = slots <= 4 * nt;
    if (few) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * nt;
            my_row[k] = i < slots ? slot_to_row(a, col, i) : 0;
            my_meta[k] = i < slots ? a.meta[my_row[k]] : 0u;
        }
    }
    // this column's candidate count (bookkeeping only)
    int cand_pre = (temporal && tid < nf) ? a.cand_cnt[(int64_t)r * nf + tid] : 0;
    if (tid == 0) ecount = 0;
    lds_barrier();                 // not __syncthreads(): that one would drain the loads just issued
    if (temporal) {
        // compaction with ONE LDS atomic per wave and round (a same-address atomic per kept edge serialises)
        for (int j0 = 0; j0 < total; j0 += PER * nt) {
            if (j0 > 0) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int j = j0 + k * nt + tid;
                    local[k] = j < total ? elist[j] : -1;
                }
            }
            int nv = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) nv += local[k] != -1 ? 1 : 0;
            int incl = nv;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            int base = 0;
            if (lane == 63 && incl) base = atomicAdd(&ecount, incl);
            base = __shfl(base, 63, 64);
            int pos = base + incl - nv;
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (local[k] != -1) cst<GMEM>(edges + pos++, local[k]);
        }
        for (int t = tid + nt; t < nf; t += nt) cand_pre += a.cand_cnt[(int64_t)r * nf + t];
    }
    // node areas (0 = no node starts at this slot): one coalesced pass over meta, reused by every later phase
    if (few) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * nt;
            if (i < slots) carea_st<GMEM>(area_l, i, my_meta[k] ? box_area(my_meta[k], my_row[k], a.H * a.W, a.W) : 0);
        }
    } else {
        for (int i = tid; i < slots; i += nt) {
            const int row = slot_to_row(a, col, i);
            const uint32_t m = a.meta[row];
            carea_st<GMEM>(area_l, i, m ? box_area(m, row, a.H * a.W, a.W) : 0);
        }
    }
    for (int i = tid; i < slots; i += nt) { cst<GMEM>(rep + i, i); cst<GMEM>(rep2 + i, i); }
    col_sync<GMEM>();
    E = ecount;
    STTM_TICK(1);

    int probe_iters = 0;
    if (MODE != COL_FINAL && temporal) {
        // ---- PROBE: iterate to the fixed point, remember after which iterations the labels were idempotent --
        unsigned long long mask = 0ull;
        int it = 0;
        bool overflow = false;
        while (true) {
            // alternate flag pairs: iteration k+1 resets the OTHER pair, so no barrier is needed between reading this
            // iteration's flags and starting the next one (the pair is reused two iterations, i.e. >= 3 barriers, later)
            int* fl = flags + 2 * (it & 1);
            column_iteration<GMEM>(rep, rep2, edges, E, slots, fl);
            const int changed = fl[0], bad = fl[1];
            if (!bad) mask |= 1ull << it;
            ++it;
            if (!changed) break;                 // fixed point: stable => idempotent from here on
            if (it >= kMaxProbeIters) { overflow = true; break; }
        }
        if (tid == 0) {
            mask |= ~0ull << (it - 1);           // the labels no longer move: every later iteration is idempotent
            __hip_atomic_store(a.col_mask + r, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (overflow) atomicAdd(a.counts + STTM_CNT_OVERFLOW, 1);
        }
        probe_iters = it;
    }
    if constexpr (MODE == COL_PROBE) return;
    STTM_TICK(2);
    if constexpr (MODE == COL_FUSED) grid_barrier(a.bar + 0, R, a.counts + STTM_CNT_OVERFLOW);
    STTM_TICK(3);
    {
        // ---- FINAL: K = first iteration after which EVERY column is idempotent; labels after exactly K iterations --
        int K = 0;
        if (temporal) {
            unsigned long long m = ~0ull;
            for (int c = tid; c < R; c += nt) m &= __hip_atomic_load(a.col_mask + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m &= __shfl_xor(m, d, 64);
            if (lane == 0) wmask[wave] = m;
            col_sync<GMEM>();
            unsigned long long all = ~0ull;
            for (int w = 0; w < nwave; ++w) all &= wmask[w];
            K = all ? __ffsll((long long)all) : kMaxProbeIters;      // lowest set bit index + 1
            // fused: this column already sits at its fixed point, reached after probe_iters - 1 iterations; that is the
            // answer whenever K is at least that.  Otherwise (and in the two-kernel path) replay exactly K iterations.
            if (MODE == COL_FINAL || K < probe_iters - 1) {
                if (MODE == COL_FUSED) {
                    for (int i = tid; i < slots; i += nt) { cst<GMEM>(rep + i, i); cst<GMEM>(rep2 + i, i); }
                    col_sync<GMEM>();
                }
                for (int it = 0; it < K; ++it) column_iteration<GMEM>(rep, rep2, edges, E, slots, flags);
            }
        }
        STTM_TICK(4);
        // from here: rep = final labels; rep2, the edge array and `mem` are free
        int* gcnt = rep2;
        const int HW = a.H * a.W;
        for (int i = tid; i < slots; i += nt) cst<GMEM>(gcnt + i, 0);
        col_sync<GMEM>();
        int nodes = 0, leafnodes = 0;
        for (int i = tid; i < slots; i += nt) {
            const int ar = carea_ld<GMEM>(area_l, i);
            if (ar) { caadd<GMEM>(gcnt + cld<GMEM>(rep + i), 1); ++nodes; leafnodes += ar == 1 ? 1 : 0; }
        }
        col_sync<GMEM>();
        STTM_TICK(5);
        // offsets of the groups inside this column's slice of `members` (exclusive scan over slots)
        {
            const int per = (slots + nt - 1) / nt;
            const int lo = tid * per < slots ? tid * per : slots, hi = lo + per < slots ? lo + per : slots;
            int mine = 0;
            for (int i = lo; i < hi; ++i) mine += cld<GMEM>(gcnt + i);
            int tot = 0;
            int off = block_exclusive_scan<!GMEM>(mine, wsum, &tot);
            for (int i = lo; i < hi; ++i) {
                const int n = cld<GMEM>(gcnt + i);
                cst<GMEM>(aux + i, off);
                const int row = slot_to_row(a, col, i);
                a.grp_off[row] = col.base + off;      // plain stores: the consumer is the next kernel
                a.grp_cnt[row] = n;                   // 0 for non-survivors: the group-mean kernel keys on this
                off += n;
            }
        }
        col_sync<GMEM>();
        STTM_TICK(6);
        // survivors per frame -> frame_cnt (one atomic per (frame, column)); the group-mean kernel turns them into row
        // prefixes.  `survivors` = this column's share of N'.
        int survivors = 0;
        if (col.A <= 64) {
            for (int t = tid; t < a.T; t += nt) {
                int c = 0;
                for (int q = 0; q < col.A; ++q) c += cld<GMEM>(gcnt + t * col.A + q) > 0 ? 1 : 0;
                if (c) atomicAdd(a.frame_cnt + t, c);
                survivors += c;
            }
        } else {
            for (int t = wave; t < a.T; t += nwave) {
                int c = 0;
                for (int q = lane; q < col.A; q += 64) c += cld<GMEM>(gcnt + t * col.A + q) > 0 ? 1 : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
                if (lane == 0 && c) atomicAdd(a.frame_cnt + t, c);
                if (lane == 0) survivors += c;
            }
        }
        STTM_TICK(7);
        // unordered fill (cursor = aux), then order every multi-member group ascending and publish global rows
        for (int i = tid; i < slots; i += nt) {
            if (carea_ld<GMEM>(area_l, i)) {
                const int pos = caadd<GMEM>(aux + cld<GMEM>(rep + i), 1);
                cst<GMEM>(mem + pos, i);
            }
        }
        col_sync<GMEM>();
        int32_t* out = a.members + col.base;
        // node-parallel ordering: every node ranks itself inside its group's (unordered) list -- O(group size) per node,
        // balanced over the threads -- and publishes its origin row at that rank; the representative totals the patches
        for (int i = tid; i < slots; i += nt) {
            const int ar = carea_ld<GMEM>(area_l, i);
            if (!ar) continue;
            const int r = cld<GMEM>(rep + i);
            const int n = cld<GMEM>(gcnt + r);
            const int o = cld<GMEM>(aux + r) - n;          // the fill cursor ended at offset + n
            int rk = 0;
            if (n > 1)
                for (int j = 0; j < n; ++j) rk += cld<GMEM>(mem + o + j) < i ? 1 : 0;
            const int row = slot_to_row(a, col, i);
            out[o + rk] = row | (ar == 1 ? kLeafBit : 0);
            if (r == i) {
                int patches = ar;
                if (n > 1) {
                    patches = 0;
                    for (int j = 0; j < n; ++j) patches += carea_ld<GMEM>(area_l, cld<GMEM>(mem + o + j));
                }
                a.grp_np[row] = patches;
            }
        }
        STTM_TICK(8);
        // ---- bookkeeping counters and N'.  The per-wave partials meet in LDS; thread 0 adds this column's totals to the
        // global counters, waits for exactly those atomics, and then adds (survivors << 24 | 1) to ONE 64-bit word
        // (R < 2^24 columns, N' < 2^31): the add that completes the arrival count also returns the complete N', so the
        // last column publishes everything to the host with no second grid barrier and no drain of the other threads'
        // stores (their consumer is the next kernel).
        int cand = temporal ? cand_pre : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            nodes += __shfl_xor(nodes, d, 64); leafnodes += __shfl_xor(leafnodes, d, 64);
            survivors += __shfl_xor(survivors, d, 64); cand += __shfl_xor(cand, d, 64);
        }
        __shared__ int part[4][16];
        if (lane == 0) { part[0][wave] = nodes; part[1][wave] = leafnodes; part[2][wave] = survivors; part[3][wave] = cand; }
        STTM_TICK(9);
        col_sync<GMEM>();
        if (tid == 0) {
            int tot[4] = {0, 0, 0, 0};
            for (int w = 0; w < nwave; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k) tot[k] += part[k][w];
            if (tot[0]) atomicAdd(a.counts + STTM_CNT_NODES, tot[0]);
            if (tot[1]) atomicAdd(a.counts + STTM_CNT_LEAFNODES, tot[1]);
            if (tot[3]) atomicAdd(a.counts + STTM_CNT_CANDIDATES, tot[3]);
            if (E) atomicAdd(a.counts + STTM_CNT_EDGES, E);
            if (r == 0) st_agent(a.counts + STTM_CNT_ITERS, K);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* word = reinterpret_cast<unsigned long long*>(a.bar + 2);
            const unsigned long long old = __hip_atomic_fetch_add(word, ((unsigned long long)(unsigned)tot[2] << 24) | 1ull,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            STTM_TICK(10);
            if ((int)(old & 0xffffffull) == R - 1) publish_counts(a, (int)(old >> 24) + tot[2]);
        }
        STTM_TICK(11);
        if (a.dbg_ticks && blockIdx.x == a.dbg_wg && threadIdx.x == 0) a.dbg_ticks[13] = clock64();
    }
}

constexpr size_t kColLdsLimit = 160 * 1024 - 1024;      // leave room for the static __shared__ scratch

// Threads per column workgroup.  The phases are short and mostly per-thread bookkeeping (scans, address arithmetic): with
// 16 waves on the CU's 4 SIMDs every instruction of a wave waits for three other waves to issue theirs.
int col_threads(const TemporalArgs& a) {
    static const int cap_env = [] { const char* e = getenv("STTM_LABEL_NT"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 0; }();
    const int cap = cap_env ? cap_env : kColThreads;
    int nthreads = 256;
    while (nthreads < cap && nthreads * 2 <= a.max_slots) nthreads *= 2;
    return nthreads;
}

bool col_labels_use_gmem(const TemporalArgs& a) {
    return a.force_gmem || (size_t)18 * a.max_slots > kColLdsLimit;
}

hipError_t launch_col_labels(const TemporalArgs& a, bool probe, hipStream_t stream) {
    const bool gmem = col_labels_use_gmem(a);
    const int nthreads = col_threads(a);
    const size_t smem = gmem ? 0 : (size_t)18 * a.max_slots;
    if (probe) {
        if (!(a.temporal_thresh > 0.f && a.T > 1)) return hipSuccess;
        if (gmem) hipLaunchKernelGGL((k_col_labels<COL_PROBE, true>), dim3(a.R), dim3(nthreads), smem, stream, a);
        else hipLaunchKernelGGL((k_col_labels<COL_PROBE, false>), dim3(a.R), dim3(nthreads), smem, stream, a);
    } else {
        if (gmem) hipLaunchKernelGGL((k_col_labels<COL_FINAL, true>), dim3(a.R), dim3(nthreads), smem, stream, a);
        else hipLaunchKernelGGL((k_col_labels<COL_FINAL, false>), dim3(a.R), dim3(nthreads), smem, stream, a);
    }
    return hipGetLastError();
}

// One launch for probe + final + rank when every column workgroup is certainly resident at once (R <= 128 CUs'
// worth, one workgroup per CU at most): the two global agreements become in-kernel grid barriers.
bool labels_can_fuse(const TemporalArgs& a) { return a.R <= 128 && !a.no_fuse; }

hipError_t launch_labels_fused(const TemporalArgs& a, hipStream_t stream) {
    const bool gmem = col_labels_use_gmem(a);
    size_t smem = gmem ? 0 : (size_t)18 * a.max_slots;
    const int nthreads = col_threads(a);
    if (gmem) hipLaunchKernelGGL((k_col_labels<COL_FUSED, true>), dim3(a.R), dim3(nthreads), smem, stream, a);
    else hipLaunchKernelGGL((k_col_labels<COL_FUSED, false>), dim3(a.R), dim3(nthreads), smem, stream, a);
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
// the member count (or the patch count when weighted_avg).
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

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_group_mean(TemporalArgs a) {
    // Workgroup (t, s): `gm_split` workgroups share frame t.  Every wave scans the frame's H*W origin slots (one ballot
    // per 64 slots gives each survivor its rank inside the frame; rows of earlier frames are the sum of frame_cnt before it) and takes
    // the survivors whose in-frame rank is congruent to its id, so the output order is (frame, y1, x1) with no rank pass.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int S = a.gm_split, t = blockIdx.x / S, s = blockIdx.x - t * S;
    const int HW = a.H * a.W;
    const int stride = S * nwave, me = s * nwave + wave;
    // output rows before this frame: every wave sums frame_cnt[0..t) for itself (no LDS, no workgroup barrier)
    int row0 = 0;
    for (int f = lane; f < t; f += 64) row0 += a.frame_cnt[f];
    int j0 = 0;
    bool row0_done = false;
    constexpr int NB = 4;                                       // 64-slot chunks whose metadata is loaded together
    for (int base = 0; base < HW; base += 64 * NB) {
        int cnt[NB], np[NB], go[NB];
        uint32_t meta[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {                          // independent loads, one round trip
            const int p = base + b * 64 + lane;
            const bool in = p < HW;
            const int origin = t * HW + (in ? p : 0);
            cnt[b] = in ? a.grp_cnt[origin] : 0;
            np[b] = a.grp_np[origin];
            go[b] = a.grp_off[origin];
            meta[b] = a.meta[origin];
        }
        if (!row0_done) { row0 = wave_sum_int(row0); row0_done = true; }      // after the metadata loads were issued
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const unsigned long long m = __ballot(cnt[b] > 0);
            if (m == 0ull) continue;
            const int j = j0 + __popcll(m & ((1ull << lane) - 1ull));
            unsigned long long sel = __ballot(cnt[b] > 0 && (j & (stride - 1)) == me);     // stride is a power of two
            while (sel) {
                const int l = __ffsll((long long)sel) - 1;
                sel &= sel - 1ull;
                const int p = base + b * 64 + l;
                const int row = row0 + j0 + __popcll(m & ((1ull << l) - 1ull));
                const int n = __builtin_amdgcn_readlane(cnt[b], l), off = __builtin_amdgcn_readlane(go[b], l);
                const int patches = __builtin_amdgcn_readlane(np[b], l);
                const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)meta[b], l);
                const int origin = t * HW + p;
                const int y1 = p / a.W, x1 = p - y1 * a.W;
                const int y2 = (int)(mt >> 16), x2 = (int)(mt & 0xffff);
                const bool leaf = (y2 - y1) == 1 && (x2 - x1) == 1;
                if (a.npatch_out && lane == 0) {
                    a.npatch_out[row] = patches;
                    int32_t* o = a.tlbr_out + (int64_t)row * 5;
                    o[0] = t; o[1] = y1; o[2] = x1; o[3] = y2; o[4] = x2;
                }
                const void* s0 = (leaf && a.xrows) ? a.xrows : a.S;          // 1x1 nodes were not copied out of x
                const bool divide = a.weighted_avg || n > 1;
                const float den = round_to<T>(a.weighted_avg ? (float)patches : (float)n);
                // U chunks of the row in flight per lane (wide rows would otherwise be a chain of load -> store round trips);
                // per element the members are still added in ascending order, one rounding per add
                constexpr int U = TypeInfo<T>::lowp ? 4 : 2;
                for (int cb = lane * VEC; cb < a.C; cb += U * 64 * VEC) {
                    Pack<T, VEC> acc[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int c0 = cb + u * 64 * VEC;
                        if (c0 < a.C) acc[u] = load_pack<T, VEC>(s0, (int64_t)origin * a.C + c0); else acc[u].zero();
                    }
                    for (int k = 1; k < n; ++k) {
                        const int mr = a.members[off + k];
                        const void* sm = (mr < 0 && a.xrows) ? a.xrows : a.S;
                        const int64_t mbase = (int64_t)(mr & 0x7fffffff) * a.C;
                        Pack<T, VEC> q[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int c0 = cb + u * 64 * VEC;
                            if (c0 < a.C) q[u] = load_pack<T, VEC>(sm, mbase + c0); else q[u].zero();
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const Pack<T, VEC> prev = acc[u];
                            pack_fill(acc[u], [&](int e) { return prev.get(e) + q[u].get(e); });
                        }
                    }
                    if (divide) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const Pack<T, VEC> prev = acc[u];
                            pack_fill(acc[u], [&](int e) { return prev.get(e) / den; });
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int c0 = cb + u * 64 * VEC;
                        if (c0 < a.C) store_pack_stream<T, VEC>(a.feat_out, (int64_t)row * a.C + c0, acc[u]);
                    }
                }
            }
            j0 += __popcll(m);
        }
    }
}

hipError_t launch_group_mean(const TemporalArgs& a, hipStream_t stream) {
    const int grid = a.T * a.gm_split;
#define STTM_LAUNCH_GM(TT, VV) hipLaunchKernelGGL((k_group_mean<TT, VV>), dim3(grid), dim3(256), 0, stream, a)
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
