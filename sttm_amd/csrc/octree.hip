// Octree baseline of replace_qwen2_by_sparse_attn: replaces octree_build / get_octree_features
// (token_merging_utils/octree_utils.py:293-389 of the reference) for the whole cubes of a clip.
//
// A cube = `side` consecutive frames of side x side tokens.  Per axis the pyramid follows the quadtree's rule (an odd side
// keeps its first cell alone, :17-148), a parent is compared with its 8 child slots (missing slots alias child cell (0,0,0) of
// the same cube, :204-205 / :262-266) and is emitted whole when all 8 cosines reach the threshold (:281-289); emitted
// nodes are ordered by the leaf index of their first corner (:369-373).
//
// Level by level, every kernel HBM-bound and without host synchronisation:
//   k_oct_pool    one thread per (parent cell, 16-byte pack): float32 sum of the 1/2/4/8 children in (t, y, x) order, one rounding
//   k_oct_stop    one wave per parent cell: parent row + 8 child rows in ONE pass -> 8 dots, 9 squared norms -> stop flag
//                 (the stop decision of a cell does not depend on the frontier, so every level is decided in full)
//   k_oct_emit    one thread per leaf: walks its ancestors top-down to the first stopped one; the leaf that is that node's
//                 first corner marks the node (origin addressing: the marks are already in output order)
//   k_oct_scan_*  exclusive sum of the marks = output rows (4096 marks per workgroup, then the workgroup offsets), N' to the caller's
//                 device counter
//   k_oct_gather  one wave per marked leaf: copies the node's row from its pyramid level
#include "sttm_kernels.h"

namespace sttm {

__device__ __forceinline__ int oct_parent(int c, int n_child) {          // inverse of child_start / child_count
    return (n_child & 1) ? (c == 0 ? 0 : (c + 1) / 2) : c / 2;
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_oct_pool(const void* child, void* parent, int B, int sc, int sp, int C) {
#pragma clang fp contract(off)
    const int P = C / VEC;
    const int64_t total = (int64_t)B * sp * sp * sp * P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int pk = (int)(idx % P);
        int64_t cell = idx / P;
        const int k = (int)(cell % sp), j = (int)((cell / sp) % sp), i = (int)((cell / ((int64_t)sp * sp)) % sp);
        const int b = (int)(cell / ((int64_t)sp * sp * sp));
        const int ts = child_start(i, sc), tc = child_count(i, sc);
        const int ys = child_start(j, sc), yc = child_count(j, sc);
        const int xs = child_start(k, sc), xc = child_count(k, sc);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int dt = 0; dt < tc; ++dt)
            for (int dy = 0; dy < yc; ++dy)
                for (int dx = 0; dx < xc; ++dx) {
                    const int64_t row = (((int64_t)b * sc + ts + dt) * sc + ys + dy) * sc + xs + dx;
                    const Pack<T, VEC> p = load_pack<T, VEC>(child, row * C + pk * VEC);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = acc[e] + p.get(e);
                }
        const float den = (float)(tc * yc * xc);
        Pack<T, VEC> res;
#pragma unroll
        for (int e = 0; e < VEC; ++e) res.set(e, acc[e] / den);
        store_pack<T, VEC>(parent, cell * C + pk * VEC, res);
    }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_oct_stop(const void* parent, const void* child, uint8_t* stop, int B, int sp, int sc, int C,
                                                  double thr_lo_sq) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int64_t cells = (int64_t)B * sp * sp * sp;
    for (int64_t cell = (int64_t)blockIdx.x * nwave + wave; cell < cells; cell += (int64_t)gridDim.x * nwave) {
        const int k = (int)(cell % sp), j = (int)((cell / sp) % sp), i = (int)((cell / ((int64_t)sp * sp)) % sp);
        const int b = (int)(cell / ((int64_t)sp * sp * sp));
        const int ts = child_start(i, sc), tc = child_count(i, sc);
        const int ys = child_start(j, sc), yc = child_count(j, sc);
        const int xs = child_start(k, sc), xc = child_count(k, sc);
        int64_t crow[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int dt = s >> 2, dy = (s >> 1) & 1, dx = s & 1;
            const bool ok = dt < tc && dy < yc && dx < xc;
            // a slot without a child aliases child cell (0, 0, 0) of the same cube
            crow[s] = ok ? (((int64_t)b * sc + ts + dt) * sc + ys + dy) * sc + xs + dx : (int64_t)b * sc * sc * sc;
        }
        float st[17];                                        // 8 dots, 8 child norms^2, parent norm^2
#pragma unroll
        for (int q = 0; q < 17; ++q) st[q] = 0.f;
        for (int c0 = lane * VEC; c0 < C; c0 += 64 * VEC) {
            const Pack<T, VEC> p = load_pack<T, VEC>(parent, cell * C + c0);
            Pack<T, VEC> ch[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) ch[s] = load_pack<T, VEC>(child, crow[s] * C + c0);
            st[16] += dot_pack(p, p);
#pragma unroll
            for (int s = 0; s < 8; ++s) { st[s] += dot_pack(p, ch[s]); st[8 + s] += dot_pack(ch[s], ch[s]); }
        }
#pragma unroll
        for (int q = 0; q < 17; ++q) st[q] = wave_sum(st[q]);
        if (lane == 0) {
            bool all = true;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const double a = fmax((double)st[16], 1e-16), bb = fmax((double)st[8 + s], 1e-16);
                const double d = (double)st[s];
                all = all && (d * fabs(d) >= thr_lo_sq * (a * bb));
            }
            stop[cell] = all ? 1 : 0;
        }
    }
}

__global__ void __launch_bounds__(256) k_oct_emit(OctArgs a) {
    const int S = a.side[a.L - 1];
    const int64_t leaves = (int64_t)a.B * S * S * S;
    for (int64_t leaf = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; leaf < leaves; leaf += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(leaf % S), y = (int)((leaf / S) % S), t = (int)((leaf / ((int64_t)S * S)) % S);
        const int b = (int)(leaf / ((int64_t)S * S * S));
        // ancestor cell per level, bottom-up
        int ct[kOctMaxLevels], cy[kOctMaxLevels], cx[kOctMaxLevels];
        ct[a.L - 1] = t; cy[a.L - 1] = y; cx[a.L - 1] = x;
        for (int l = a.L - 2; l >= 0; --l) {
            ct[l] = oct_parent(ct[l + 1], a.side[l + 1]);
            cy[l] = oct_parent(cy[l + 1], a.side[l + 1]);
            cx[l] = oct_parent(cx[l + 1], a.side[l + 1]);
        }
        // first stopped ancestor, top-down (the leaf level always stops)
        int e = a.L - 1;
        for (int l = 0; l < a.L - 1; ++l) {
            const int n = a.side[l];
            if (a.stop[l][(((int64_t)b * n + ct[l]) * n + cy[l]) * n + cx[l]]) { e = l; break; }
        }
        // is this leaf the first corner of that node?
        int ft = ct[e], fy = cy[e], fx = cx[e];
        for (int l = e; l < a.L - 1; ++l) {
            ft = child_start(ft, a.side[l + 1]); fy = child_start(fy, a.side[l + 1]); fx = child_start(fx, a.side[l + 1]);
        }
        const bool origin = ft == t && fy == y && fx == x;
        a.mark[leaf] = origin ? 1 : 0;
        a.level_of[leaf] = (uint8_t)e;
    }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_oct_gather(OctArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int S = a.side[a.L - 1];
    const int64_t leaves = (int64_t)a.B * S * S * S;
    for (int64_t leaf = (int64_t)blockIdx.x * nwave + wave; leaf < leaves; leaf += (int64_t)gridDim.x * nwave) {
        if (!a.mark[leaf]) continue;
        const int e = a.level_of[leaf];
        int ct = (int)((leaf / ((int64_t)S * S)) % S), cy = (int)((leaf / S) % S), cx = (int)(leaf % S);
        const int b = (int)(leaf / ((int64_t)S * S * S));
        for (int l = a.L - 2; l >= e; --l) {
            ct = oct_parent(ct, a.side[l + 1]); cy = oct_parent(cy, a.side[l + 1]); cx = oct_parent(cx, a.side[l + 1]);
        }
        const int n = a.side[e];
        const int64_t src = (((int64_t)b * n + ct) * n + cy) * n + cx;
        const int64_t row = a.rows[leaf];
        constexpr int U = 4;                                  // chunks of the row in flight per lane
        for (int cb = lane * VEC; cb < a.C; cb += U * 64 * VEC) {
            Pack<T, VEC> v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c0 = cb + u * 64 * VEC;
                if (c0 < a.C) v[u] = load_pack<T, VEC>(a.feat[e], src * a.C + c0); else v[u].zero();
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c0 = cb + u * 64 * VEC;
                if (c0 < a.C) store_pack<T, VEC>(a.out, row * a.C + c0, v[u]);
            }
        }
    }
}

// Exclusive sum of the leaf marks (round 6: in-tree, it was hipcub::DeviceScan + a one-thread count kernel).  Pass 1: a workgroup scans
// 4096 marks (four per thread, wave prefix by lane shifts, wave totals through LDS) and leaves its total in blocksum[b]; pass 2: workgroup b
// adds the sum of the totals before it (a few hundred at most: one strided read + a workgroup reduction) and the last one writes N'.
constexpr int OS_T = 1024, OS_PER = 4, OS_BLK = OS_T * OS_PER;
__global__ void __launch_bounds__(OS_T) k_oct_scan_local(const int32_t* __restrict__ mark, int32_t* __restrict__ rows, int64_t n,
                                                         int32_t* __restrict__ blocksum) {
    __shared__ int wsum[OS_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * OS_BLK + (int64_t)tid * OS_PER;
    int v[OS_PER];
#pragma unroll
    for (int k = 0; k < OS_PER; ++k) v[k] = base + k < n ? mark[base + k] : 0;
    int total = 0;
#pragma unroll
    for (int k = 0; k < OS_PER; ++k) total += v[k];
    int inc = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int pre = inc - total;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
#pragma unroll
    for (int k = 0; k < OS_PER; ++k) {
        if (base + k < n) rows[base + k] = pre;
        pre += v[k];
    }
    if (tid == OS_T - 1) blocksum[blockIdx.x] = pre;
}

__global__ void __launch_bounds__(OS_T) k_oct_scan_add(const int32_t* __restrict__ mark, int32_t* __restrict__ rows, int64_t n,
                                                       const int32_t* __restrict__ blocksum, int32_t* __restrict__ count_out) {
    __shared__ int wsum[OS_T / 64];
    __shared__ int off_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int part = 0;
    for (int b = tid; b < (int)blockIdx.x; b += OS_T) part += blocksum[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int w = 0; w < OS_T / 64; ++w) o += wsum[w];
        off_sh = o;
    }
    __syncthreads();
    const int off = off_sh;
    const int64_t base = (int64_t)blockIdx.x * OS_BLK + (int64_t)tid * OS_PER;
#pragma unroll
    for (int k = 0; k < OS_PER; ++k) {
        if (base + k < n) {
            const int rw = rows[base + k] + off;
            if (off) rows[base + k] = rw;
            if (base + k == n - 1) *count_out = rw + mark[n - 1];
        }
    }
}

size_t octree_scan_bytes(int64_t n) { return (size_t)((n + OS_BLK - 1) / OS_BLK) * sizeof(int32_t); }

template <typename T>
static hipError_t octree_run_t(OctArgs& a, int vec, void* scan_tmp, size_t scan_bytes, hipStream_t stream) {
    auto grid_for = [](int64_t threads) { int64_t g = (threads + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); };
#define STTM_OCT_VEC(KERNEL, ...)                                                                                     \
    do {                                                                                                              \
        if constexpr (TypeInfo<T>::lowp) {                                                                            \
            if (vec == 8) hipLaunchKernelGGL((KERNEL<T, 8>), __VA_ARGS__);                                            \
            else if (vec == 4) hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__);                                       \
            else hipLaunchKernelGGL((KERNEL<T, 2>), __VA_ARGS__);                                                     \
        } else {                                                                                                      \
            if (vec == 4) hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__);                                            \
            else if (vec == 2) hipLaunchKernelGGL((KERNEL<T, 2>), __VA_ARGS__);                                       \
            else hipLaunchKernelGGL((KERNEL<T, 1>), __VA_ARGS__);                                                     \
        }                                                                                                             \
    } while (0)
    for (int l = a.L - 2; l >= 0; --l) {                                  // pyramid, fine to coarse
        const int sp = a.side[l], sc = a.side[l + 1];
        const int64_t threads = (int64_t)a.B * sp * sp * sp * (a.C / vec);
        STTM_OCT_VEC(k_oct_pool, dim3(grid_for(threads)), dim3(256), 0, stream, a.feat[l + 1], const_cast<void*>(a.feat[l]), a.B, sc, sp, a.C);
    }
    for (int l = 0; l < a.L - 1; ++l) {
        const int sp = a.side[l], sc = a.side[l + 1];
        const int64_t cells = (int64_t)a.B * sp * sp * sp;
        STTM_OCT_VEC(k_oct_stop, dim3(grid_for(cells * 64)), dim3(256), 0, stream, a.feat[l], a.feat[l + 1], a.stop[l], a.B, sp, sc, a.C,
                     a.thr_lo_sq);
    }
    const int S = a.side[a.L - 1];
    const int64_t leaves = (int64_t)a.B * S * S * S;
    hipLaunchKernelGGL(k_oct_emit, dim3(grid_for(leaves)), dim3(256), 0, stream, a);
    const unsigned scan_blocks = (unsigned)((leaves + OS_BLK - 1) / OS_BLK);
    if ((size_t)scan_blocks * sizeof(int32_t) > scan_bytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_oct_scan_local, dim3(scan_blocks), dim3(OS_T), 0, stream, a.mark, a.rows, leaves, reinterpret_cast<int32_t*>(scan_tmp));
    hipLaunchKernelGGL(k_oct_scan_add, dim3(scan_blocks), dim3(OS_T), 0, stream, a.mark, a.rows, leaves, reinterpret_cast<const int32_t*>(scan_tmp), a.count_out);
    STTM_OCT_VEC(k_oct_gather, dim3(grid_for(leaves * 64)), dim3(256), 0, stream, a);
#undef STTM_OCT_VEC
    return hipGetLastError();
}

hipError_t launch_octree(OctArgs& a, int dtype, int vec, void* scan_tmp, size_t scan_bytes, hipStream_t stream) {
    if (dtype == STTM_F32) return octree_run_t<float>(a, vec, scan_tmp, scan_bytes, stream);
    if (dtype == STTM_BF16) return octree_run_t<bf16_t>(a, vec, scan_tmp, scan_bytes, stream);
    return octree_run_t<f16_t>(a, vec, scan_tmp, scan_bytes, stream);
}

}  // namespace sttm
