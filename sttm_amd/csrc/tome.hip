// K7/K8 -- ToMe bipartite soft matching + size-weighted merge on gfx950 (tome_token_merger.py:13-91).
//
// One sttm_tome_step call = one iteration of tome_per_video's loop (tome_token_merger.py:143-149):
//   k_tome_normalize   head-mean metric, unit rows (no eps), split into even (a) / odd (b) token matrices
//   k_tome_match       scores = a @ b^T fused with the row max / argmax -- the [na, nb] score matrix
//                      (629 MB at n = 25 088) is never materialised.  fp32-input MFMA
//                      (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation) computes the
//                      TRANSPOSED tile D[j][i] = b_j . a_i, so every lane owns one a-row i and sees its
//                      candidates j along its accumulator registers: the running max/argmax needs no
//                      cross-lane traffic inside the main loop.
//   hipcub radix sort  ranking of the a-tokens by best score, descending (stable)
//   k_tome_*           per-destination source lists in rank order, then the size-weighted merge
// Nothing here depends on data-dependent sizes: the step is enqueued without any host synchronisation.
#include <hipcub/hipcub.hpp>

#include "sttm_kernels.h"

namespace sttm {

// ---------------------------------------------------------------------------------------------------
// normalise: m = mean over heads of x.reshape(n, n_head, D); m /= |m| ; a = m[0::2], b = m[1::2]
// one wave per token row
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tome_normalize(const float* __restrict__ x, int n, int C, int n_head, int D, int Dp,
                                                        float* __restrict__ ahat, float* __restrict__ bhat) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int row = blockIdx.x * nwave + wave; row < n; row += gridDim.x * nwave) {
        const float* xr = x + (int64_t)row * C;
        float* out = ((row & 1) ? bhat : ahat) + (int64_t)(row >> 1) * Dp;   // rows padded with zeros to the k-tile
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) {
            float m;
            if (n_head == 1) {
                m = xr[d];
            } else {
                float s = 0.f;
                for (int h = 0; h < n_head; ++h) s += xr[h * D + d];
                m = s / (float)n_head;
            }
            out[d] = m;
            ss = fmaf(m, m, ss);
        }
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
        for (int d = lane; d < D; d += 64) out[d] = out[d] / nrm;
        for (int d = D + lane; d < Dp; d += 64) out[d] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------
// match: for every a-row i: max_j <a_i, b_j> and the first j that attains it
// ---------------------------------------------------------------------------------------------------
constexpr int TM_I = 128;      // a-rows per workgroup
constexpr int TM_J = 128;      // b-rows per step
constexpr int TM_K = 32;       // k-depth per LDS stage
constexpr int TM_LD = TM_I + 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned long long pack_score(float v, int j) {
    // order-preserving map of the float, then "smaller j wins" on ties (torch.max returns the first maximum)
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)j);
}

__global__ void __launch_bounds__(256, 2) k_tome_match(const float* __restrict__ ahat, const float* __restrict__ bhat,
                                                        int na, int nb, int D, int jsplit,
                                                        unsigned long long* __restrict__ best /*[na]*/) {
    __shared__ float As[TM_K][TM_LD];     // As[k][i]
    __shared__ float Bs[TM_K][TM_LD];     // Bs[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;          // wave tile: 64 (j) x 64 (i)
    const int itile = blockIdx.x / jsplit, jpart = blockIdx.x % jsplit;
    const int i0 = itile * TM_I;
    const int jtiles = (nb + TM_J - 1) / TM_J;
    const int jt_lo = (int)((long long)jtiles * jpart / jsplit), jt_hi = (int)((long long)jtiles * (jpart + 1) / jsplit);

    // staging map: 8 threads per row (8 x float4 = 32 k), 32 rows per pass, 4 passes per 128-row tile
    const int srow = tid >> 3, skq = (tid & 7) * 4;

    float bestv[2] = {-INFINITY, -INFINITY};
    int bestj[2] = {0x7fffffff, 0x7fffffff};
    const int lcol = lane & 31, lhalf = lane >> 5;

    // Software pipeline: the global loads of k-tile n+1 are issued right after tile n has been written to LDS, so their
    // latency is covered by the 64 MFMAs of tile n instead of stalling the workgroup at the top of every k step.
    float4 ra[4], rb[4];
    // rows are padded to a multiple of TM_K (zeros) by k_tome_normalize, so a stage never needs a k bound; rows past
    // the end are clamped to the last row: their scores are computed and then ignored (j < nb / i < na below)
    auto fetch = [&](int j0, int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ri = min(i0 + p * 32 + srow, na - 1), rj = min(j0 + p * 32 + srow, nb - 1);
            ra[p] = *reinterpret_cast<const float4*>(ahat + (int64_t)ri * D + k0 + skq);
            rb[p] = *reinterpret_cast<const float4*>(bhat + (int64_t)rj * D + k0 + skq);
        }
    };
    if (jt_lo < jt_hi) fetch(jt_lo * TM_J, 0);

    for (int jt = jt_lo; jt < jt_hi; ++jt) {
        const int j0 = jt * TM_J;
        f32x16 acc[2][2];     // [j-subtile][i-subtile]
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][q][e] = 0.f;

        for (int k0 = 0; k0 < D; k0 += TM_K) {
            __syncthreads();          // previous stage fully consumed
            // registers -> LDS transposed [k][row]
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = p * 32 + srow;
                As[skq + 0][r] = ra[p].x; As[skq + 1][r] = ra[p].y; As[skq + 2][r] = ra[p].z; As[skq + 3][r] = ra[p].w;
                Bs[skq + 0][r] = rb[p].x; Bs[skq + 1][r] = rb[p].y; Bs[skq + 2][r] = rb[p].z; Bs[skq + 3][r] = rb[p].w;
            }
            __syncthreads();
            if (k0 + TM_K < D) fetch(j0, k0 + TM_K);
            else if (jt + 1 < jt_hi) fetch(j0 + TM_J, 0);
            // MFMA A operand = b rows (j), B operand = a rows (i):  D[j][i] += sum_k b[j][k] * a[i][k]
            // operands of step kk+2 are read from LDS while the four MFMAs of step kk run
            float fb[2][2], fa[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) fb[0][p] = Bs[lhalf][wj * 64 + p * 32 + lcol];
#pragma unroll
            for (int q = 0; q < 2; ++q) fa[0][q] = As[lhalf][wi * 64 + q * 32 + lcol];
#pragma unroll
            for (int kk = 0; kk < TM_K; kk += 2) {
                const int cur = (kk >> 1) & 1;
                if (kk + 2 < TM_K) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) fb[cur ^ 1][p] = Bs[kk + 2 + lhalf][wj * 64 + p * 32 + lcol];
#pragma unroll
                    for (int q = 0; q < 2; ++q) fa[cur ^ 1][q] = As[kk + 2 + lhalf][wi * 64 + q * 32 + lcol];
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of the MFMAs they overlap with
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][p], fa[cur][q], acc[p][q], 0, 0, 0);
            }
        }
        // running max over this tile: lane owns column i = wi*64 + q*32 + lcol; rows j = (e&3) + 8*(e>>2) + 4*lhalf
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = j0 + wj * 64 + p * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                    const float v = acc[p][q][e];
                    if (j < nb && (v > bestv[q] || (v == bestv[q] && j < bestj[q]))) { bestv[q] = v; bestj[q] = j; }
                }
    }
    // publish: packed 64-bit max per a-row (combines the two lane halves, the two j-waves and the j-splits)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = i0 + wi * 64 + q * 32 + lcol;
        if (i < na && bestj[q] != 0x7fffffff) atomicMax(best + i, pack_score(bestv[q], bestj[q]));
    }
}


// ---------------------------------------------------------------------------------------------------
// 16-bit inputs (bfloat16 / float16): the reference runs the same torch ops on the hidden states' dtype
// (tome_attn_monkey_patch.py:88-107 with a bf16 model), so every intermediate tensor is ROUNDED to that dtype:
// the unit rows, the scores (an fp32-accumulating matmul whose result is stored in 16 bits -- hence many exact ties,
// first maximum = smaller j), x * size, every scatter-add step, the final division.  Matrix products run on
// v_mfma_f32_32x32x16_{bf16,f16} (16x the fp32-input rate).
// ---------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float tome_round(float f);
template <> __device__ __forceinline__ float tome_round<float>(float f) { return f; }
template <> __device__ __forceinline__ float tome_round<bf16_t>(float f) { return bf16_bits_to_float(float_to_bf16_bits(f)); }
template <> __device__ __forceinline__ float tome_round<f16_t>(float f) { return f16_bits_to_float(float_to_f16_bits(f)); }
template <typename T> __device__ __forceinline__ float tome_ld(const void* p, int64_t i);
template <> __device__ __forceinline__ float tome_ld<float>(const void* p, int64_t i) { return reinterpret_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float tome_ld<bf16_t>(const void* p, int64_t i) { return bf16_bits_to_float(reinterpret_cast<const uint16_t*>(p)[i]); }
template <> __device__ __forceinline__ float tome_ld<f16_t>(const void* p, int64_t i) { return f16_bits_to_float(reinterpret_cast<const uint16_t*>(p)[i]); }
template <typename T> __device__ __forceinline__ void tome_st(void* p, int64_t i, float v);
template <> __device__ __forceinline__ void tome_st<float>(void* p, int64_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
template <> __device__ __forceinline__ void tome_st<bf16_t>(void* p, int64_t i, float v) { reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)float_to_bf16_bits(v); }
template <> __device__ __forceinline__ void tome_st<f16_t>(void* p, int64_t i, float v) { reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)float_to_f16_bits(v); }

// unit rows in the input dtype: m = mean over heads (fp32 sum, rounded), |m| = sqrt(fp32 sum of squares) rounded,
// m / |m| rounded; rows padded with zeros to the 32-wide k tile.  One wave per token row.
template <typename T>
__global__ void __launch_bounds__(256) k_tome_normalize16(const void* __restrict__ x, int n, int C, int n_head, int D, int Dp,
                                                          uint16_t* __restrict__ ahat, uint16_t* __restrict__ bhat) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int row = blockIdx.x * nwave + wave; row < n; row += gridDim.x * nwave) {
        uint16_t* out = ((row & 1) ? bhat : ahat) + (int64_t)(row >> 1) * Dp;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) {
            float m;
            if (n_head == 1) {
                m = tome_ld<T>(x, (int64_t)row * C + d);
            } else {
                float s = 0.f;
                for (int h = 0; h < n_head; ++h) s += tome_ld<T>(x, (int64_t)row * C + h * D + d);
                m = tome_round<T>(s / (float)n_head);
            }
            tome_st<T>(out, d, m);
            ss = fmaf(m, m, ss);
        }
        ss = wave_sum(ss);
        const float nrm = tome_round<T>(sqrtf(ss));
        for (int d = lane; d < D; d += 64) tome_st<T>(out, d, tome_ld<T>(out, d) / nrm);
        for (int d = D + lane; d < Dp; d += 64) out[d] = 0;
    }
}

typedef __bf16 tome_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tome_f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <typename T> struct TomeMfma;
template <> struct TomeMfma<bf16_t> {
    typedef tome_bf16x8 vec;
    static __device__ __forceinline__ f32x16_t run(vec a, vec b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct TomeMfma<f16_t> {
    typedef tome_f16x8 vec;
    static __device__ __forceinline__ f32x16_t run(vec a, vec b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// Same structure as k_tome_match (transposed tile D[j][i] = b_j . a_i, lane = one a-row, running max in registers, packed
// atomicMax), on 16-bit operands: LDS tiles are [row][k] (what the 32x32x16 operand wants: 8 consecutive k per lane, one
// ds_read_b128), rows padded to 80 bytes so that the 16 lanes of a ds_read_b128 group hit 16 different 4-bank groups.
constexpr int TM16_LD = TM_K + 8;      // elements per LDS row
template <typename T>
__global__ void __launch_bounds__(256, 2) k_tome_match16(const uint16_t* __restrict__ ahat, const uint16_t* __restrict__ bhat,
                                                          int na, int nb, int D, int jsplit,
                                                          unsigned long long* __restrict__ best /*[na]*/) {
    __shared__ __attribute__((aligned(16))) uint16_t As[TM_I * TM16_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[TM_J * TM16_LD];
    typedef typename TomeMfma<T>::vec vec;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;          // wave tile: 64 (j) x 64 (i)
    const int itile = blockIdx.x / jsplit, jpart = blockIdx.x % jsplit;
    const int i0 = itile * TM_I;
    const int jtiles = (nb + TM_J - 1) / TM_J;
    const int jt_lo = (int)((long long)jtiles * jpart / jsplit), jt_hi = (int)((long long)jtiles * (jpart + 1) / jsplit);
    // staging map: 4 threads per row (4 x 16 bytes = 32 k), 64 rows per pass, 2 passes per 128-row tile
    const int srow = tid >> 2, sch = (tid & 3) * 8;
    float bestv[2] = {-INFINITY, -INFINITY};
    int bestj[2] = {0x7fffffff, 0x7fffffff};
    const int lcol = lane & 31, lhalf = lane >> 5;
    uint4 ra[2], rb[2];
    auto fetch = [&](int j0, int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ri = min(i0 + p * 64 + srow, na - 1), rj = min(j0 + p * 64 + srow, nb - 1);
            ra[p] = *reinterpret_cast<const uint4*>(ahat + (int64_t)ri * D + k0 + sch);
            rb[p] = *reinterpret_cast<const uint4*>(bhat + (int64_t)rj * D + k0 + sch);
        }
    };
    if (jt_lo < jt_hi) fetch(jt_lo * TM_J, 0);
    for (int jt = jt_lo; jt < jt_hi; ++jt) {
        const int j0 = jt * TM_J;
        f32x16_t acc[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][q][e] = 0.f;
        for (int k0 = 0; k0 < D; k0 += TM_K) {
            __syncthreads();          // previous stage fully consumed
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                *reinterpret_cast<uint4*>(As + (p * 64 + srow) * TM16_LD + sch) = ra[p];
                *reinterpret_cast<uint4*>(Bs + (p * 64 + srow) * TM16_LD + sch) = rb[p];
            }
            __syncthreads();
            if (k0 + TM_K < D) fetch(j0, k0 + TM_K);
            else if (jt + 1 < jt_hi) fetch(j0 + TM_J, 0);
#pragma unroll
            for (int ks = 0; ks < TM_K; ks += 16) {
                vec fb[2], fa[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) fb[p] = *reinterpret_cast<const vec*>(Bs + (wj * 64 + p * 32 + lcol) * TM16_LD + ks + lhalf * 8);
#pragma unroll
                for (int q = 0; q < 2; ++q) fa[q] = *reinterpret_cast<const vec*>(As + (wi * 64 + q * 32 + lcol) * TM16_LD + ks + lhalf * 8);
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[p][q] = TomeMfma<T>::run(fb[p], fa[q], acc[p][q]);
            }
        }
        // the score tensor of the reference has the input dtype: round before comparing (ties -> smaller j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = j0 + wj * 64 + p * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                    const float v = tome_round<T>(acc[p][q][e]);
                    if (j < nb && (v > bestv[q] || (v == bestv[q] && j < bestj[q]))) { bestv[q] = v; bestj[q] = j; }
                }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = i0 + wi * 64 + q * 32 + lcol;
        if (i < na && bestj[q] != 0x7fffffff) atomicMax(best + i, pack_score(bestv[q], bestj[q]));
    }
}

__global__ void k_tome_unpack(const unsigned long long* __restrict__ best, int na, float* __restrict__ node_max,
                              int* __restrict__ node_idx, int* __restrict__ iota) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    const unsigned long long b = best[i];
    unsigned u = (unsigned)(b >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    node_max[i] = b ? __uint_as_float(u) : __uint_as_float(0x7fc00000u);       // no finite score at all: NaN row
    node_idx[i] = b ? (int)(0xffffffffu - (unsigned)(b & 0xffffffffu)) : 0;
    iota[i] = i;
}

// ---------------------------------------------------------------------------------------------------
// merge bookkeeping: sources (rank < r) grouped per destination b-token, in rank order
// ---------------------------------------------------------------------------------------------------
__global__ void k_tome_count(const int* __restrict__ order, const int* __restrict__ node_idx, int r, int* __restrict__ cnt) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < r) atomicAdd(cnt + node_idx[order[k]], 1);
}

__global__ void __launch_bounds__(1024) k_tome_scan(const int* __restrict__ cnt, int nb, int* __restrict__ off) {
    // single workgroup exclusive scan (nb is a few 10^4 at most)
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += nt) {
        const int i = base + tid;
        const int v = i < nb ? cnt[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (i < nb) off[i] = pre + inc - v;
        __syncthreads();
        if (tid == nt - 1) carry = pre + inc;
        __syncthreads();
    }
    if (tid == 0) off[nb] = carry;
}

__global__ void k_tome_fill(const int* __restrict__ order, const int* __restrict__ node_idx, int r, const int* __restrict__ off,
                            int* __restrict__ cur, int* __restrict__ lists /* rank positions k */) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < r) {
        const int d = node_idx[order[k]];
        lists[off[d] + atomicAdd(cur + d, 1)] = k;
    }
}

// one wave per output row.  rows [0, na - r): unmerged a-tokens in rank order; rows [na - r, n - r): b-tokens.
// Every tensor of the reference is rounded to the input dtype T (x * size, the scatter-added sums, the sums of sizes, the
// quotient): for T = float that is the plain fp32 arithmetic of the reference, without contraction.
template <typename T>
__global__ void __launch_bounds__(256) k_tome_merge(const void* __restrict__ x, const float* __restrict__ size,
                                                    const int64_t* __restrict__ idx, int n, int C, int na, int nb, int r,
                                                    const int* __restrict__ order, const int* __restrict__ off,
                                                    int* __restrict__ lists, void* __restrict__ x_out,
                                                    float* __restrict__ size_out, int64_t* __restrict__ idx_out) {
#pragma clang fp contract(off)      // x*size is rounded before it is added, like the reference's mul then scatter-add

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int n_out = n - r;
    for (int row = blockIdx.x * nwave + wave; row < n_out; row += gridDim.x * nwave) {
        if (row < na - r) {
            const int tok = 2 * order[r + row];                   // unmerged a-token: (x*size)/size
            const float s = size ? size[tok] : 1.f;
            for (int c = lane; c < C; c += 64)
                tome_st<T>(x_out, (int64_t)row * C + c, tome_round<T>(tome_ld<T>(x, (int64_t)tok * C + c) * s) / s);
            if (lane == 0) { size_out[row] = s; idx_out[row] = idx[tok]; }
            continue;
        }
        const int j = row - (na - r);
        const int tok = 2 * j + 1;
        const int o = off[j], cnt = off[j + 1] - o;
        // order this destination's sources by rank (ascending k)
        if (cnt > 1) {
            if (cnt <= 64) {                                      // the usual case: rank sort in registers
                const int v = lane < cnt ? lists[o + lane] : 0x7fffffff;
                int rk = 0;
                for (int m = 0; m < cnt; ++m) rk += __shfl(v, m, 64) < v ? 1 : 0;
                if (lane < cnt) lists[o + rk] = v;
            } else {                                              // pathological inputs: selection sort, O(cnt^2 / 64)
                for (int m = 0; m < cnt - 1; ++m) {
                    int mn = 0x7fffffff, mp = -1;
                    for (int q = m + lane; q < cnt; q += 64) { const int v = lists[o + q]; if (v < mn) { mn = v; mp = q; } }
                    for (int d = 32; d >= 1; d >>= 1) {
                        const int omn = __shfl_xor(mn, d, 64), omp = __shfl_xor(mp, d, 64);
                        if (omn < mn) { mn = omn; mp = omp; }
                    }
                    if (lane == 0 && mp != m) { const int tmp = lists[o + m]; lists[o + m] = mn; lists[o + mp] = tmp; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        // scatter_add accumulates in fp32 and rounds the sum ONCE to the tensor dtype (ATen's expanded-index path, checked on
        // bf16 against the CPU operator); the products x * size are tensors of their own and are rounded individually
        const float sb = size ? size[tok] : 1.f;
        float stot = sb;
        for (int m = 0; m < cnt; ++m) {
            const int atok = 2 * order[lists[o + m]];
            stot = stot + (size ? size[atok] : 1.f);
        }
        stot = tome_round<T>(stot);
        for (int c = lane; c < C; c += 64) {
            float acc = tome_round<T>(tome_ld<T>(x, (int64_t)tok * C + c) * sb);     // plain operators: the __fmul_rn/__fadd_rn wrappers fuse once inlined
            for (int m = 0; m < cnt; ++m) {
                const int atok = 2 * order[lists[o + m]];
                const float sa = size ? size[atok] : 1.f;
                acc = acc + tome_round<T>(tome_ld<T>(x, (int64_t)atok * C + c) * sa);
            }
            tome_st<T>(x_out, (int64_t)row * C + c, tome_round<T>(acc) / stot);
        }
        if (lane == 0) { size_out[row] = stot; idx_out[row] = idx[tok]; }
    }
}

struct TomePlan {
    int na, nb, D, Dp;
    size_t off_ahat, off_bhat, off_best, off_nmax, off_nidx, off_iota, off_keys, off_order, off_cnt, off_cur, off_off,
        off_lists, off_cub, cub_bytes, total;
};

static int tome_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

static int tome_plan(int n, int C, int n_head, TomePlan* p) {
    if (n < 2 || C < 1 || n_head < 1 || C % n_head) return -1;
    p->na = (n + 1) / 2; p->nb = n / 2; p->D = C / n_head;
    p->Dp = (p->D + TM_K - 1) / TM_K * TM_K;
    size_t o = 0;
    p->off_ahat = o; o = al(o + (size_t)p->na * p->Dp * 4);
    p->off_bhat = o; o = al(o + (size_t)(p->nb > 0 ? p->nb : 1) * p->Dp * 4);
    p->off_best = o; o = al(o + (size_t)p->na * 8);
    p->off_nmax = o; o = al(o + (size_t)p->na * 4);
    p->off_nidx = o; o = al(o + (size_t)p->na * 4);
    p->off_iota = o; o = al(o + (size_t)p->na * 4);
    p->off_keys = o; o = al(o + (size_t)p->na * 4);
    p->off_order = o; o = al(o + (size_t)p->na * 4);
    p->off_cnt = o; o = al(o + (size_t)(p->nb + 1) * 4);
    p->off_cur = o; o = al(o + (size_t)(p->nb + 1) * 4);
    p->off_off = o; o = al(o + (size_t)(p->nb + 2) * 4);
    p->off_lists = o; o = al(o + (size_t)p->na * 4);
    size_t cub = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                                 (int*)nullptr, p->na);
    p->cub_bytes = cub;
    p->off_cub = o; o = al(o + cub);
    p->total = o;
    return 0;
}

}  // namespace sttm

extern "C" {

size_t sttm_tome_workspace_bytes(int n, int C, int n_head) {
    sttm::TomePlan p;
    if (sttm::tome_plan(n, C, n_head, &p) != 0) return 0;
    return p.total;
}

int sttm_tome_step(const void* x_, const float* size, const int64_t* idx, int n, int C, int n_head, int r, int dtype,
                   void* workspace, size_t workspace_bytes, void* x_out_, float* size_out, int64_t* idx_out,
                   float* node_max_out, int32_t* node_idx_out, void* stream_) {
    using namespace sttm;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (dtype < 0 || dtype > 2) return STTM_ERR_ARG;
    if (!x_ || !idx || !workspace || !x_out_ || !size_out || !idx_out) return STTM_ERR_ARG;
    TomePlan p;
    if (tome_plan(n, C, n_head, &p) != 0) return STTM_ERR_ARG;
    if (workspace_bytes < p.total) return STTM_ERR_ARG;
    if (r < 1 || r > p.nb) return STTM_ERR_ARG;              // callers clamp r = min(r, n // 2) like the reference
    char* ws = reinterpret_cast<char*>(workspace);
    float* ahat = reinterpret_cast<float*>(ws + p.off_ahat);       // (16-bit inputs keep 16-bit unit rows in the same buffers)
    float* bhat = reinterpret_cast<float*>(ws + p.off_bhat);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ws + p.off_best);
    float* nmax = reinterpret_cast<float*>(ws + p.off_nmax);
    int* nidx = reinterpret_cast<int*>(ws + p.off_nidx);
    int* iota = reinterpret_cast<int*>(ws + p.off_iota);
    float* keys = reinterpret_cast<float*>(ws + p.off_keys);
    int* order = reinterpret_cast<int*>(ws + p.off_order);
    int* cnt = reinterpret_cast<int*>(ws + p.off_cnt);
    int* cur = reinterpret_cast<int*>(ws + p.off_cur);
    int* off = reinterpret_cast<int*>(ws + p.off_off);
    int* lists = reinterpret_cast<int*>(ws + p.off_lists);

    (void)hipMemsetAsync(best, 0, (size_t)p.na * 8, stream);
    (void)hipMemsetAsync(cnt, 0, p.off_off - p.off_cnt, stream);      // cnt and cur
    const int ngrid = [&] { int g = (n + 3) / 4; return g > 8192 ? 8192 : g; }();
    const int itiles = (p.na + TM_I - 1) / TM_I;
    const int jtiles = (p.nb + TM_J - 1) / TM_J;
    // j-split: the grid is itiles*jsplit workgroups of ceil(jtiles/jsplit) tile products each, two resident per CU.
    // Pick the split with the smallest per-CU critical path  ceil(WGs / CUs) * tiles-per-WG  (a 588-WG grid on 512
    // slots leaves a third of the chip idle for the second round); ties go to the coarser split (fewer atomics).
    const int n_cu = tome_cu_count();
    int jsplit = 1;
    long best_cost = -1;
    for (int js = 1; js <= jtiles; ++js) {
        const long wgs = (long)itiles * js;
        const long per_cu = (wgs + n_cu - 1) / n_cu;
        long cost = per_cu * ((jtiles + js - 1) / js);
        if (per_cu < 2 && js < jtiles) cost = cost * 3 / 2;       // a lone WG per CU cannot hide its staging
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; jsplit = js; }
    }
    if (dtype == STTM_F32) {
        hipLaunchKernelGGL(k_tome_normalize, dim3(ngrid), dim3(256), 0, stream, reinterpret_cast<const float*>(x_), n, C, n_head, p.D, p.Dp, ahat, bhat);
        hipLaunchKernelGGL(k_tome_match, dim3(itiles * jsplit), dim3(256), 0, stream, ahat, bhat, p.na, p.nb, p.Dp, jsplit, best);
    } else if (dtype == STTM_BF16) {
        hipLaunchKernelGGL(k_tome_normalize16<bf16_t>, dim3(ngrid), dim3(256), 0, stream, x_, n, C, n_head, p.D, p.Dp,
                           reinterpret_cast<uint16_t*>(ahat), reinterpret_cast<uint16_t*>(bhat));
        hipLaunchKernelGGL(k_tome_match16<bf16_t>, dim3(itiles * jsplit), dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(ahat),
                           reinterpret_cast<const uint16_t*>(bhat), p.na, p.nb, p.Dp, jsplit, best);
    } else {
        hipLaunchKernelGGL(k_tome_normalize16<f16_t>, dim3(ngrid), dim3(256), 0, stream, x_, n, C, n_head, p.D, p.Dp,
                           reinterpret_cast<uint16_t*>(ahat), reinterpret_cast<uint16_t*>(bhat));
        hipLaunchKernelGGL(k_tome_match16<f16_t>, dim3(itiles * jsplit), dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(ahat),
                           reinterpret_cast<const uint16_t*>(bhat), p.na, p.nb, p.Dp, jsplit, best);
    }
    hipLaunchKernelGGL(k_tome_unpack, dim3((p.na + 255) / 256), dim3(256), 0, stream, best, p.na, nmax, nidx, iota);
    size_t cub = p.cub_bytes;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(ws + p.off_cub, cub, nmax, keys, iota, order, p.na, 0, 32, stream);
    hipLaunchKernelGGL(k_tome_count, dim3((r + 255) / 256), dim3(256), 0, stream, order, nidx, r, cnt);
    hipLaunchKernelGGL(k_tome_scan, dim3(1), dim3(1024), 0, stream, cnt, p.nb, off);
    hipLaunchKernelGGL(k_tome_fill, dim3((r + 255) / 256), dim3(256), 0, stream, order, nidx, r, off, cur, lists);
    {
        int grid = (n - r + 3) / 4; if (grid > 8192) grid = 8192;
#define STTM_TOME_MERGE(TT) hipLaunchKernelGGL(k_tome_merge<TT>, dim3(grid), dim3(256), 0, stream, x_, size, idx, n, C, p.na, p.nb, r, order, off, lists, x_out_, size_out, idx_out)
        if (dtype == STTM_F32) STTM_TOME_MERGE(float); else if (dtype == STTM_BF16) STTM_TOME_MERGE(bf16_t); else STTM_TOME_MERGE(f16_t);
#undef STTM_TOME_MERGE
    }
    if (node_max_out) (void)hipMemcpyAsync(node_max_out, nmax, (size_t)p.na * 4, hipMemcpyDeviceToDevice, stream);
    if (node_idx_out) (void)hipMemcpyAsync(node_idx_out, nidx, (size_t)p.na * 4, hipMemcpyDeviceToDevice, stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STTM_OK : STTM_ERR_LAUNCH;
}

}  // extern "C"
