// DyCoke stage-1 temporal token pruning (the "dycoke-stage1" baseline of replace_qwen2_by_sparse_attn):
// replaces dycoke_ttm (token_merging_utils/dycoke_merger.py:8-83 of the reference).
//
//   pass 1 (:13-45)  frames (2j, 2j+1): per-token cosine; frame 2j+1 keeps its k LEAST similar tokens, in ascending
//                    similarity order (topk(largest=False)); frame 2j stays whole.  An odd last frame stays whole (:47-52).
//   pass 2 (:54-79)  frames (4j, 4j+2) with 4j < T-4: frame 4j+2 (whole so far) keeps its k least similar tokens w.r.t. 4j.
//   output           frames in order, kept tokens + their flat token ids (:81-83).
//
// Every output size is known in advance (k = int((1 - prune_ratio) * P) per pruned frame), so the three kernels are enqueued
// with no host synchronisation:
//   k_dycoke_sim     one wave per (frame pair, token): F.cosine_similarity's normalise-then-dot form (float32; k_dycoke_sim16
//                    for bfloat16 / float16 with the per-op rounding of the input dtype)
//   k_dycoke_select  one workgroup per frame pair: bitonic sort of (similarity, token) ascending in LDS, first k tokens
//   k_dycoke_gather  one wave per output row: copies the row, writes its token id
// HBM-bound: the similarity kernel reads every frame once for pass 1 and half of them again for pass 2.
#include "sttm_kernels.h"

namespace sttm {

struct DycokeArgs {
    const float* x;         // [T*P, C]
    int T, P, C, k, n1, n2;
    float* sim;             // [n1+n2][P]
    int32_t* keep;          // [n1+n2][k]
    float* out;             // [rows, C]
    int64_t* out_idx;       // [rows]
};

__device__ __forceinline__ void dycoke_pair(const DycokeArgs& a, int j, int& fa, int& fb) {
    if (j < a.n1) { fa = 2 * j; fb = fa + 1; } else { fa = 4 * (j - a.n1); fb = fa + 2; }
}
// pair that prunes frame f, or -1 when the frame stays whole
__host__ __device__ __forceinline__ int dycoke_pruner(int f, int T, int n1) {
    if (f & 1) return (f - 1) / 2;                          // pass 1: every odd frame
    if ((f & 3) == 2 && f - 2 < T - 4) return n1 + (f - 2) / 4;   // pass 2
    return -1;
}

// NV = 16-byte chunks per lane and row kept in registers (C <= 256 * NV, C % 4 == 0); NV == 0: any C, rows re-read from cache
template <int NV>
__global__ void __launch_bounds__(256) k_dycoke_sim(DycokeArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int64_t total = (int64_t)(a.n1 + a.n2) * a.P;
    for (int64_t w = (int64_t)blockIdx.x * nwave + wave; w < total; w += (int64_t)gridDim.x * nwave) {
        const int j = (int)(w / a.P), p = (int)(w - (int64_t)j * a.P);
        int fa, fb;
        dycoke_pair(a, j, fa, fb);
        const float* ra = a.x + ((int64_t)fa * a.P + p) * a.C;
        const float* rb = a.x + ((int64_t)fb * a.P + p) * a.C;
        // F.cosine_similarity (ATen): x / max(|x|, eps) on both sides, then the sum of products
        float sa = 0.f, sb = 0.f, d = 0.f;
        if constexpr (NV > 0) {
            float4 u[NV], v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                u[i] = c < a.C ? *reinterpret_cast<const float4*>(ra + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[i] = c < a.C ? *reinterpret_cast<const float4*>(rb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                sa = fmaf(u[i].x, u[i].x, sa); sa = fmaf(u[i].y, u[i].y, sa); sa = fmaf(u[i].z, u[i].z, sa); sa = fmaf(u[i].w, u[i].w, sa);
                sb = fmaf(v[i].x, v[i].x, sb); sb = fmaf(v[i].y, v[i].y, sb); sb = fmaf(v[i].z, v[i].z, sb); sb = fmaf(v[i].w, v[i].w, sb);
            }
            sa = wave_sum(sa); sb = wave_sum(sb);
            const float na = fmaxf(sqrtf(sa), 1e-8f), nb = fmaxf(sqrtf(sb), 1e-8f);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                d = fmaf(u[i].x / na, v[i].x / nb, d); d = fmaf(u[i].y / na, v[i].y / nb, d);
                d = fmaf(u[i].z / na, v[i].z / nb, d); d = fmaf(u[i].w / na, v[i].w / nb, d);
            }
        } else {
            for (int c = lane; c < a.C; c += 64) { const float u = ra[c], v = rb[c]; sa = fmaf(u, u, sa); sb = fmaf(v, v, sb); }
            sa = wave_sum(sa); sb = wave_sum(sb);
            const float na = fmaxf(sqrtf(sa), 1e-8f), nb = fmaxf(sqrtf(sb), 1e-8f);
            for (int c = lane; c < a.C; c += 64) d = fmaf(ra[c] / na, rb[c] / nb, d);
        }
        d = wave_sum(d);
        if (lane == 0) a.sim[w] = d;
    }
}

// 16-bit inputs (bfloat16 / float16 hidden states): F.cosine_similarity runs on the input dtype, i.e. every intermediate tensor
// is rounded to it (checked against the ATen CPU result, tests/golden/make_golden_dycoke16.py): |a| = round(sqrt(sum a^2)),
// a / max(|a|, eps) rounded, the products rounded, their fp32 sum rounded.  One wave per (frame pair, token); the row is
// read twice (norms, then the dot), the second time from cache.
template <typename T> __device__ __forceinline__ float dyc_round(float f);
template <> __device__ __forceinline__ float dyc_round<bf16_t>(float f) { return bf16_bits_to_float(float_to_bf16_bits(f)); }
template <> __device__ __forceinline__ float dyc_round<f16_t>(float f) { return f16_bits_to_float(float_to_f16_bits(f)); }
template <typename T> __device__ __forceinline__ float dyc_cvt(uint32_t bits16);
template <> __device__ __forceinline__ float dyc_cvt<bf16_t>(uint32_t b) { return bf16_bits_to_float(b); }
template <> __device__ __forceinline__ float dyc_cvt<f16_t>(uint32_t b) { return f16_bits_to_float(b); }

template <typename T, bool VEC8>
__global__ void __launch_bounds__(256) k_dycoke_sim16(DycokeArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int64_t total = (int64_t)(a.n1 + a.n2) * a.P;
    const uint16_t* x = reinterpret_cast<const uint16_t*>(a.x);
    for (int64_t w = (int64_t)blockIdx.x * nwave + wave; w < total; w += (int64_t)gridDim.x * nwave) {
        const int j = (int)(w / a.P), p = (int)(w - (int64_t)j * a.P);
        int fa, fb;
        dycoke_pair(a, j, fa, fb);
        const uint16_t* ra = x + ((int64_t)fa * a.P + p) * a.C;
        const uint16_t* rb = x + ((int64_t)fb * a.P + p) * a.C;
        float sa = 0.f, sb = 0.f, d = 0.f;
        if constexpr (VEC8) {
            for (int c = lane * 8; c < a.C; c += 512) {
                const uint4 u = *reinterpret_cast<const uint4*>(ra + c), v = *reinterpret_cast<const uint4*>(rb + c);
                const uint32_t uw[4] = {u.x, u.y, u.z, u.w}, vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u0 = dyc_cvt<T>(uw[i] & 0xffffu), u1 = dyc_cvt<T>(uw[i] >> 16);
                    const float v0 = dyc_cvt<T>(vw[i] & 0xffffu), v1 = dyc_cvt<T>(vw[i] >> 16);
                    sa = fmaf(u0, u0, sa); sa = fmaf(u1, u1, sa); sb = fmaf(v0, v0, sb); sb = fmaf(v1, v1, sb);
                }
            }
        } else {
            for (int c = lane; c < a.C; c += 64) { const float u = dyc_cvt<T>(ra[c]), v = dyc_cvt<T>(rb[c]); sa = fmaf(u, u, sa); sb = fmaf(v, v, sb); }
        }
        sa = wave_sum(sa); sb = wave_sum(sb);
        const float na = fmaxf(dyc_round<T>(sqrtf(sa)), 1e-8f), nb = fmaxf(dyc_round<T>(sqrtf(sb)), 1e-8f);
        auto term = [&](float u, float v) { return dyc_round<T>(dyc_round<T>(u / na) * dyc_round<T>(v / nb)); };
        if constexpr (VEC8) {
            for (int c = lane * 8; c < a.C; c += 512) {
                const uint4 u = *reinterpret_cast<const uint4*>(ra + c), v = *reinterpret_cast<const uint4*>(rb + c);
                const uint32_t uw[4] = {u.x, u.y, u.z, u.w}, vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d += term(dyc_cvt<T>(uw[i] & 0xffffu), dyc_cvt<T>(vw[i] & 0xffffu));
                    d += term(dyc_cvt<T>(uw[i] >> 16), dyc_cvt<T>(vw[i] >> 16));
                }
            }
        } else {
            for (int c = lane; c < a.C; c += 64) d += term(dyc_cvt<T>(ra[c]), dyc_cvt<T>(rb[c]));
        }
        d = wave_sum(d);
        if (lane == 0) a.sim[w] = dyc_round<T>(d);
    }
}

__global__ void __launch_bounds__(1024) k_dycoke_select(DycokeArgs a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* key = reinterpret_cast<float*>(smem_raw);
    int* val = reinterpret_cast<int*>(key + npad);
    const int j = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < npad; i += nt) {
        key[i] = i < a.P ? a.sim[(int64_t)j * a.P + i] : INFINITY;
        val[i] = i < a.P ? i : 0x7fffffff;
    }
    __syncthreads();
    // ascending by (similarity, token): ties go to the smaller token id (torch.topk leaves their order unspecified)
    for (int size = 2; size <= npad; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < npad; i += nt) {
                const int partner = i ^ stride;
                if (partner > i) {
                    const bool up = (i & size) == 0;
                    const float ki = key[i], kp = key[partner];
                    const int vi = val[i], vp = val[partner];
                    const bool gt = ki > kp || (ki == kp && vi > vp);
                    if (gt == up) { key[i] = kp; key[partner] = ki; val[i] = vp; val[partner] = vi; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < a.k; i += nt) a.keep[(int64_t)j * a.k + i] = val[i];
}

template <int NV>
__global__ void __launch_bounds__(256) k_dycoke_gather(DycokeArgs a, int split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int f = blockIdx.x / split, s = blockIdx.x - f * split;
    // rows before frame f, closed form: the odd frames below f and the pass-2 frames 4j+2 < f (j < n2) hold k rows each
    const int c2 = (f + 1) / 4 < a.n2 ? (f + 1) / 4 : a.n2;
    const int pruned_before = f / 2 + c2;
    const int64_t row0 = (int64_t)pruned_before * a.k + (int64_t)(f - pruned_before) * a.P;
    const int pr = dycoke_pruner(f, a.T, a.n1);
    const int rows = pr >= 0 ? a.k : a.P;
    for (int r = s * nwave + wave; r < rows; r += split * nwave) {
        const int p = pr >= 0 ? a.keep[(int64_t)pr * a.k + r] : r;
        const int64_t tok = (int64_t)f * a.P + p;
        const float* __restrict__ src = a.x + tok * a.C;
        float* __restrict__ dst = a.out + (row0 + r) * a.C;
        if constexpr (NV > 0) {
            float4 v[NV];                                   // the whole row of this lane in flight at once
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                v[i] = c < a.C ? *reinterpret_cast<const float4*>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < a.C) *reinterpret_cast<float4*>(dst + c) = v[i];
            }
        } else {
            for (int c = lane; c < a.C; c += 64) dst[c] = src[c];
        }
        if (lane == 0) a.out_idx[row0 + r] = tok;
    }
}

hipError_t launch_dycoke(const void* x, int T, int P, int C, int dtype, int k, float* sim, int32_t* keep, void* out, int64_t* out_idx,
                         hipStream_t stream) {
    DycokeArgs a;
    a.x = reinterpret_cast<const float*>(x); a.T = T; a.P = P; a.C = C; a.k = k;
    a.n1 = T / 2;
    a.n2 = T > 4 ? (T - 4 + 3) / 4 : 0;
    a.sim = sim; a.keep = keep; a.out = reinterpret_cast<float*>(out); a.out_idx = out_idx;
    const int np = a.n1 + a.n2;
    if (np > 0) {
        int64_t blocks = ((int64_t)np * P + 3) / 4;
        if (blocks > 16384) blocks = 16384;
        if (dtype == STTM_F32) {
            const bool vec_ok = (C % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
            if (vec_ok && C <= 1024) hipLaunchKernelGGL(k_dycoke_sim<4>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
            else if (vec_ok && C <= 2048) hipLaunchKernelGGL(k_dycoke_sim<8>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
            else if (vec_ok && C <= 4096) hipLaunchKernelGGL(k_dycoke_sim<16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
            else hipLaunchKernelGGL(k_dycoke_sim<0>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        } else {
            const bool v8 = (C % 8 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
            if (dtype == STTM_BF16) {
                if (v8) hipLaunchKernelGGL((k_dycoke_sim16<bf16_t, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((k_dycoke_sim16<bf16_t, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
            } else {
                if (v8) hipLaunchKernelGGL((k_dycoke_sim16<f16_t, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((k_dycoke_sim16<f16_t, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
            }
        }
        if (k > 0) {
            int npad = 1;
            while (npad < P) npad <<= 1;
            const int nt = npad < 1024 ? (npad < 64 ? 64 : npad) : 1024;
            hipLaunchKernelGGL(k_dycoke_select, dim3(np), dim3(nt), (size_t)npad * 8, stream, a, npad);
        }
    }
    // the gather copies whole rows: 16-bit rows (even C) are moved as C / 2 four-byte words
    DycokeArgs g = a;
    if (dtype != STTM_F32) g.C = C / 2;
    int split = (4096 + T - 1) / T;
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    const bool vec4 = (g.C % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (vec4 && g.C <= 1024) hipLaunchKernelGGL(k_dycoke_gather<4>, dim3(T * split), dim3(256), 0, stream, g, split);
    else if (vec4 && g.C <= 2048) hipLaunchKernelGGL(k_dycoke_gather<8>, dim3(T * split), dim3(256), 0, stream, g, split);
    else if (vec4 && g.C <= 4096) hipLaunchKernelGGL(k_dycoke_gather<16>, dim3(T * split), dim3(256), 0, stream, g, split);
    else hipLaunchKernelGGL(k_dycoke_gather<0>, dim3(T * split), dim3(256), 0, stream, g, split);
    return hipGetLastError();
}

}  // namespace sttm
