// K2..K5 -- temporal stage of STTM on gfx950.
//
//   k_pairs       candidate pairs + cosine filter   (quadtree_temporal_merger.py:8-73 of the reference)
//   k_labels      synchronous hook + pointer-jump label propagation, survivor scan, group lists
//                                                   (:223-269 and the bookkeeping half of :123-171)
//   k_group_mean  per-survivor ascending-order accumulation and mean                 (:123-171)
//
// All of them address nodes by their ORIGIN ROW  t*H*W + y1*W + x1  in the scratch matrix S written by
// the spatial kernel.  Origin rows are ordered exactly like the reference's sorted node indices, so
// min-label propagation over origin rows is the same computation as over node indices.
//
// Candidate pairs never cross root cells (a node lies inside exactly one root cell and root cells are the
// same in every frame), so one workgroup per (frame pair, root cell) enumerates <= 16x16 box tests instead
// of the reference's dense [T-1, M, M, 4] tensor.
#include "sttm_kernels.h"

namespace sttm {

__device__ __forceinline__ int ld_agent(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int32_t* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// K2: pairs
// ---------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_pairs(TemporalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* cand = reinterpret_cast<int*>(smem_raw);     // [2 * cap] (index into A list, index into B list)
    __shared__ int ncand;
    const int R = a.R;
    const int t = blockIdx.x / R, r = blockIdx.x % R;
    const int HW = a.H * a.W;
    const int* LA = a.rc_list + (int64_t)(t * R + r) * a.rc_stride;
    const int* LB = a.rc_list + (int64_t)((t + 1) * R + r) * a.rc_stride;
    const int nA = LA[0], nB = LB[0];
    const int cap = 2 * (a.rc_stride - 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    if (tid == 0) ncand = 0;
    __syncthreads();
    for (int p = tid; p < nA * nB; p += blockDim.x) {
        const int ia = p / nB, ib = p - ia * nB;
        const unsigned ba = (unsigned)LA[1 + ia], bb = (unsigned)LB[1 + ib];
        const int ay1 = ba >> 24, ax1 = (ba >> 16) & 255, ay2 = (ba >> 8) & 255, ax2 = ba & 255;
        const int by1 = bb >> 24, bx1 = (bb >> 16) & 255, by2 = (bb >> 8) & 255, bx2 = bb & 255;
        const bool a_has_b = ay1 <= by1 && ax1 <= bx1 && ay2 >= by2 && ax2 >= bx2;
        const bool b_has_a = ay1 >= by1 && ax1 >= bx1 && ay2 <= by2 && ax2 <= bx2;
        if (a_has_b || b_has_a) {
            const int pos = atomicAdd(&ncand, 1);
            if (pos < cap) { cand[2 * pos] = ia; cand[2 * pos + 1] = ib; }
        }
    }
    __syncthreads();
    const int nc = ncand < cap ? ncand : cap;
    if (tid == 0) {
        atomicAdd(a.counts + STTM_CNT_CANDIDATES, ncand);
        if (ncand > cap) atomicAdd(a.counts + STTM_CNT_OVERFLOW, 1);
    }
    for (int c = wave; c < nc; c += nwave) {
        const unsigned ba = (unsigned)LA[1 + cand[2 * c]], bb = (unsigned)LB[1 + cand[2 * c + 1]];
        const int rowA = t * HW + (int)(ba >> 24) * a.W + (int)((ba >> 16) & 255);
        const int rowB = (t + 1) * HW + (int)(bb >> 24) * a.W + (int)((bb >> 16) & 255);
        float dot = 0.f;
        for (int c0 = lane * VEC; c0 < a.C; c0 += 64 * VEC) {
            const Pack<T, VEC> pa = load_pack<T, VEC>(a.S, (int64_t)rowA * a.C + c0);
            const Pack<T, VEC> pb = load_pack<T, VEC>(a.S, (int64_t)rowB * a.C + c0);
            dot += dot_pack(pa, pb);
        }
        dot = wave_sum(dot);
        if (lane == 0) {
            // x / (|x| + 1e-8) on both sides (quadtree_temporal_merger.py:62-63), evaluated in double
            const double na = sqrt((double)a.nrm2[rowA]) + 1e-8;
            const double nb = sqrt((double)a.nrm2[rowB]) + 1e-8;
            const float sim = (float)((double)dot / (na * nb));
            if (sim >= a.temporal_thresh) {
                const int e = atomicAdd(a.counts + STTM_CNT_EDGES, 1);
                if (e < a.edge_cap) { a.edges[2 * e] = rowA; a.edges[2 * e + 1] = rowB; }
                else atomicAdd(a.counts + STTM_CNT_OVERFLOW, 1);
            }
        }
    }
}

hipError_t launch_pairs(const TemporalArgs& a, hipStream_t stream) {
    if (a.T < 2) return hipSuccess;
    const int grid = (a.T - 1) * a.R;
    const size_t smem = sizeof(int) * 4 * (size_t)(a.rc_stride - 1);
#define STTM_LAUNCH_PAIRS(TT, VV) hipLaunchKernelGGL((k_pairs<TT, VV>), dim3(grid), dim3(256), smem, stream, a)
    if (a.dtype == STTM_F32) {
        if (a.vec == 4) STTM_LAUNCH_PAIRS(float, 4); else if (a.vec == 2) STTM_LAUNCH_PAIRS(float, 2); else STTM_LAUNCH_PAIRS(float, 1);
    } else if (a.dtype == STTM_BF16) {
        if (a.vec == 8) STTM_LAUNCH_PAIRS(bf16_t, 8); else if (a.vec == 4) STTM_LAUNCH_PAIRS(bf16_t, 4); else STTM_LAUNCH_PAIRS(bf16_t, 2);
    } else {
        if (a.vec == 8) STTM_LAUNCH_PAIRS(f16_t, 8); else if (a.vec == 4) STTM_LAUNCH_PAIRS(f16_t, 4); else STTM_LAUNCH_PAIRS(f16_t, 2);
    }
#undef STTM_LAUNCH_PAIRS
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// K3: labels (single workgroup).  Every access to the label arrays is an agent-scope (L2-served)
// load/store/atomic, so phases separated by __syncthreads() see each other's writes without L1 games.
// ---------------------------------------------------------------------------------------------------
constexpr int kLabelThreads = 1024;

// Block-wide exclusive scan of one int per thread; returns the exclusive prefix, *total gets the sum.
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds_wave /*[16]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds_wave[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nwave; ++w) {
        const int s = lds_wave[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// The synchronous iteration of get_merge_dst_idx_safe.  `cur`/`nxt` ping-pong; returns the buffer that
// holds the converged labels.  All threads of the (single) workgroup call it.
__device__ int32_t* propagate_labels(const int32_t* edges, int L, int N, int32_t* cur, int32_t* nxt, int32_t* emin,
                                     int* lds_flag, int* iters_out) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int iters = 0;
    while (true) {
        // m_e = min(rep[d_e], rep[s_e]) for every edge, all from the OLD labels
        for (int e = tid; e < L; e += nt) {
            const int d = edges[2 * e], s = edges[2 * e + 1];
            const int rd = ld_agent(cur + d), rs = ld_agent(cur + s);
            emin[e] = rd < rs ? rd : rs;
        }
        if (tid == 0) *lds_flag = 0;
        __syncthreads();
        // scatter-amin on both endpoints
        for (int e = tid; e < L; e += nt) {
            const int m = emin[e];
            __hip_atomic_fetch_min(cur + edges[2 * e], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min(cur + edges[2 * e + 1], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // pointer jump: nxt = cur[cur]
        for (int i = tid; i < N; i += nt) st_agent(nxt + i, ld_agent(cur + ld_agent(cur + i)));
        __syncthreads();
        // converged when nxt == nxt[nxt]
        int bad = 0;
        for (int i = tid; i < N; i += nt) {
            const int v = ld_agent(nxt + i);
            if (ld_agent(nxt + v) != v) bad = 1;
        }
        if (bad) *lds_flag = 1;
        __syncthreads();
        const int again = *lds_flag;
        __syncthreads();
        int32_t* tmp = cur; cur = nxt; nxt = tmp;
        ++iters;
        if (!again) break;
    }
    *iters_out = iters;
    return cur;
}

__global__ void __launch_bounds__(kLabelThreads) k_label_edges(const int32_t* pairs, int L, int N, int32_t* rep_out,
                                                               int32_t* rep2, int32_t* emin, int32_t* iters_out) {
    __shared__ int flag;
    for (int i = threadIdx.x; i < N; i += blockDim.x) st_agent(rep_out + i, i);
    __syncthreads();
    int iters = 0;
    int32_t* fin = propagate_labels(pairs, L, N, rep_out, rep2, emin, &flag, &iters);
    if (fin != rep_out) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) st_agent(rep_out + i, ld_agent(fin + i));
    }
    if (threadIdx.x == 0 && iters_out) *iters_out = iters;
}

hipError_t launch_label_edges(const int32_t* pairs, int L, int N, int32_t* rep_out, int32_t* rep2, int32_t* emin,
                              int32_t* iters_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_label_edges, dim3(1), dim3(kLabelThreads), 0, stream, pairs, L, N, rep_out, rep2, emin, iters_out);
    return hipGetLastError();
}

__device__ __forceinline__ int box_area(uint32_t meta, int row, int HW, int W) {
    const int y2 = meta >> 16, x2 = meta & 0xffff;
    const int rem = row % HW;
    const int y1 = rem / W, x1 = rem - y1 * W;
    return (y2 - y1) * (x2 - x1);
}

__global__ void __launch_bounds__(kLabelThreads) k_labels(TemporalArgs a) {
    __shared__ int flag;
    __shared__ int wsum[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = a.T * a.H * a.W;
    int32_t* rep = a.rep;
    for (int i = tid; i < N; i += nt) st_agent(a.rep + i, i);
    __syncthreads();
    int L = 0;
    if (a.temporal_thresh > 0.f) {
        L = ld_agent(a.counts + STTM_CNT_EDGES);
        if (L > a.edge_cap) L = a.edge_cap;
    }
    int32_t* spare = a.rep2;
    if (L > 0) {
        int iters = 0;
        rep = propagate_labels(a.edges, L, N, a.rep, a.rep2, a.emin, &flag, &iters);
        spare = (rep == a.rep) ? a.rep2 : a.rep;
        if (tid == 0) a.counts[STTM_CNT_ITERS] = iters;
    }
    // ---- survivors: nodes that are their own representative, ranked in origin-row order -----------------
    const int per = (N + nt - 1) / nt;
    const int lo = tid * per < N ? tid * per : N;
    const int hi = lo + per < N ? lo + per : N;
    int mine = 0;
    for (int i = lo; i < hi; ++i) mine += (a.meta[i] != 0u && ld_agent(rep + i) == i) ? 1 : 0;
    int n_out = 0;
    int base = block_exclusive_scan(mine, wsum, &n_out);
    for (int i = lo; i < hi; ++i) {
        if (a.meta[i] != 0u && ld_agent(rep + i) == i) {
            st_agent(a.row2origin + base, i);
            st_agent(a.rank_of + i, base);
            ++base;
        }
    }
    if (tid == 0) a.counts[STTM_CNT_OUT] = n_out;
    __syncthreads();
    // ---- group sizes (grp_cnt / grp_cur were zeroed by the host-side memset) -----------------------------
    for (int i = tid; i < N; i += nt) {
        if (a.meta[i] != 0u) {
            const int g = ld_agent(a.rank_of + ld_agent(rep + i));
            __hip_atomic_fetch_add(a.grp_cnt + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    // ---- offsets: exclusive scan of grp_cnt[0 .. n_out) ----------------------------------------------------
    const int per2 = (n_out + nt - 1) / nt;
    const int lo2 = tid * per2 < n_out ? tid * per2 : n_out;
    const int hi2 = lo2 + per2 < n_out ? lo2 + per2 : n_out;
    int sum2 = 0;
    for (int g = lo2; g < hi2; ++g) sum2 += ld_agent(a.grp_cnt + g);
    int tot2 = 0;
    int off = block_exclusive_scan(sum2, wsum, &tot2);
    for (int g = lo2; g < hi2; ++g) {
        st_agent(a.grp_off + g, off);
        off += ld_agent(a.grp_cnt + g);
    }
    if (tid == 0) st_agent(a.grp_off + n_out, tot2);
    __syncthreads();
    // ---- fill (unordered), then order every multi-member group ascending -------------------------------------
    for (int i = tid; i < N; i += nt) {
        if (a.meta[i] != 0u) {
            const int g = ld_agent(a.rank_of + ld_agent(rep + i));
            const int pos = ld_agent(a.grp_off + g) +
                            __hip_atomic_fetch_add(a.grp_cur + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st_agent(spare + pos, i);
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
    for (int g = wave; g < n_out; g += nwave) {
        const int o = ld_agent(a.grp_off + g), n = ld_agent(a.grp_cnt + g);
        if (n == 1) {
            if (lane == 0) st_agent(a.members + o, ld_agent(spare + o));
            continue;
        }
        for (int m = lane; m < n; m += 64) {
            const int v = ld_agent(spare + o + m);
            int rk = 0;
            for (int j = 0; j < n; ++j) rk += ld_agent(spare + o + j) < v ? 1 : 0;
            st_agent(a.members + o + rk, v);
        }
    }
}

hipError_t launch_labels(const TemporalArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_labels, dim3(1), dim3(kLabelThreads), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// K5: group mean.  One wave per output row; members are accumulated in ascending origin order in the
// INPUT dtype (one rounding per add for bf16/fp16, like the reference's index_add_), then divided by
// the member count (or the patch count when weighted_avg).
// ---------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float round_to(float f);
template <> __device__ __forceinline__ float round_to<float>(float f) { return f; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float f) { return bf16_bits_to_float(float_to_bf16_bits(f)); }
template <> __device__ __forceinline__ float round_to<f16_t>(float f) { return f16_bits_to_float(float_to_f16_bits(f)); }

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_group_mean(TemporalArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int n_out = a.counts[STTM_CNT_OUT];
    const int HW = a.H * a.W;
    for (int row = blockIdx.x * nwave + wave; row < n_out; row += gridDim.x * nwave) {
        const int off = a.grp_off[row], cnt = a.grp_cnt[row];
        const int origin = a.row2origin[row];
        int patches = 0;
        for (int m = lane; m < cnt; m += 64) {
            const int mem = a.members[off + m];
            patches += box_area(a.meta[mem], mem, HW, a.W);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) patches += __shfl_xor(patches, d, 64);
        const bool divide = a.weighted_avg || cnt > 1;
        const float den = round_to<T>(a.weighted_avg ? (float)patches : (float)cnt);
        for (int c0 = lane * VEC; c0 < a.C; c0 += 64 * VEC) {
            Pack<T, VEC> acc = load_pack<T, VEC>(a.S, (int64_t)origin * a.C + c0);
            for (int m = 1; m < cnt; ++m) {
                const Pack<T, VEC> p = load_pack<T, VEC>(a.S, (int64_t)a.members[off + m] * a.C + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.set(e, acc.get(e) + p.get(e));
            }
            if (divide) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.set(e, acc.get(e) / den);
            }
            store_pack<T, VEC>(a.feat_out, (int64_t)row * a.C + c0, acc);
        }
        if (lane == 0) {
            const uint32_t meta = a.meta[origin];
            const int t = origin / HW, rem = origin - t * HW;
            const int y1 = rem / a.W, x1 = rem - y1 * a.W;
            a.npatch_out[row] = patches;
            int32_t* o = a.tlbr_out + (int64_t)row * 5;
            o[0] = t; o[1] = y1; o[2] = x1; o[3] = (int)(meta >> 16); o[4] = (int)(meta & 0xffff);
        }
    }
}

hipError_t launch_group_mean(const TemporalArgs& a, hipStream_t stream) {
    const int N = a.T * a.H * a.W;
    int grid = (N + 3) / 4;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
#define STTM_LAUNCH_GM(TT, VV) hipLaunchKernelGGL((k_group_mean<TT, VV>), dim3(grid), dim3(256), 0, stream, a)
    if (a.dtype == STTM_F32) {
        if (a.vec == 4) STTM_LAUNCH_GM(float, 4); else if (a.vec == 2) STTM_LAUNCH_GM(float, 2); else STTM_LAUNCH_GM(float, 1);
    } else if (a.dtype == STTM_BF16) {
        if (a.vec == 8) STTM_LAUNCH_GM(bf16_t, 8); else if (a.vec == 4) STTM_LAUNCH_GM(bf16_t, 4); else STTM_LAUNCH_GM(bf16_t, 2);
    } else {
        if (a.vec == 8) STTM_LAUNCH_GM(f16_t, 8); else if (a.vec == 4) STTM_LAUNCH_GM(f16_t, 4); else STTM_LAUNCH_GM(f16_t, 2);
    }
#undef STTM_LAUNCH_GM
    return hipGetLastError();
}

}  // namespace sttm
