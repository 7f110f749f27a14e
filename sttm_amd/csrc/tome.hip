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
//   k_tome_rank        ranking of the a-tokens by best score, descending, ties to the smaller index: by counting (any clip length)
//   k_tome_*           per-destination source lists in rank order, then the size-weighted merge
// Nothing here depends on data-dependent sizes: the step is enqueued without any host synchronisation.
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

// Running max / argmax of one lane's scores of a 64 (j) x 32*QI (i) wave tile: acc[p][q][e] is the score of a-row q against
// candidate  jlane + p*32 + (e&3) + 8*(e>>2)  (jlane = tile origin + wave's j offset + 4 * lane half).  A lane meets its candidates in
// ascending j -- inside a tile and from tile to tile -- so the first maximum (torch.max; tome_token_merger.py:36) is the first score
// STRICTLY above the running one: for tiles that lie inside [0, nb) that is a compare and two selects per score, no branch.
// Only the last, partial tile takes the general form (round 2's, which cost ~30 instructions and an exec-mask branch per score).
typedef float tome_f32x16 __attribute__((ext_vector_type(16)));
template <int QI, typename RoundFn, int PJ = 2>
__device__ __forceinline__ void tome_running_max(const tome_f32x16 (&acc)[PJ][QI], float (&bestv)[QI], int (&bestj)[QI], int jlane, int nb,
                                                 bool inside, RoundFn rnd) {
    static_assert(PJ != 4, "the four-wave kernel has its own running max (tome_running_max_agpr)");
    if (inside) {
#pragma unroll
        for (int q = 0; q < QI; ++q) {
            float bv = bestv[q];
            int bc = -1;
#pragma unroll
            for (int p = 0; p < PJ; ++p) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = rnd(acc[p][q][e]);
                    const bool gt = v > bv;
                    bc = gt ? p * 32 + (e & 3) + 8 * (e >> 2) : bc;
                    bv = gt ? v : bv;
                }
            }
            bestj[q] = bc >= 0 ? jlane + bc : bestj[q];
            bestv[q] = bv;
        }
    } else {
#pragma unroll
        for (int q = 0; q < QI; ++q)
#pragma unroll
            for (int p = 0; p < PJ; ++p)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = jlane + p * 32 + (e & 3) + 8 * (e >> 2);
                    const float v = rnd(acc[p][q][e]);
                    if (j < nb && (v > bestv[q] || (v == bestv[q] && j < bestj[q]))) { bestv[q] = v; bestj[q] = j; }
                }
    }
}

// ---- the two-pass running max of the 256-tile kernels (round 6) -----------------------------------------------------------------
// The sweep above costs ~5 VALU instructions per score (round to the input dtype, compare, two selects) -- 128 scores per lane and tile
// product, with both waves of a SIMD in their epilogue at the same time (nothing to overlap it with).  Rounding is monotone, so
//   max_j rnd(s_j) = rnd(max_j s_j)   and   {j : rnd(s_j) = R} = {j : s_j >= theta(R)},  theta(R) = the smallest float that rounds to R:
// pass 1 is a plain fp32 maximum (v_max3_f32: half an instruction per score), ONE rounding per row, and -- only if some lane of the wave
// improves STRICTLY on its running maximum -- pass 2 finds the first candidate at or above theta (compare + select).  Same bits as the
// sweep: a candidate wins iff its rounded score is strictly above the running maximum, ties go to the smallest j, NaN scores never win.
template <typename RT> __device__ __forceinline__ float tome_round_as(float f);
template <> __device__ __forceinline__ float tome_round_as<float>(float f) { return f; }
template <> __device__ __forceinline__ float tome_round_as<bf16_t>(float f) { return bf16_bits_to_float(float_to_bf16_bits(f)); }
template <> __device__ __forceinline__ float tome_round_as<f16_t>(float f) { return f16_bits_to_float(float_to_f16_bits(f)); }
// theta(R) for a value R of the 16-bit type: the midpoint between R and its neighbour below (in VALUE; found on the 16-bit pattern, so
// subnormals and binade boundaries need no cases), which is exact in fp32, and belongs to R iff R's pattern is even (round to nearest even)
template <typename RT> __device__ __forceinline__ float tome_tie_floor(float r) {
    if constexpr (std::is_same<RT, float>::value) {
        return r;
    } else {
        uint32_t h, mag;
        if constexpr (std::is_same<RT, bf16_t>::value) h = float_to_bf16_bits(r); else h = float_to_f16_bits(r);
        mag = h & 0x7fffu;
        const bool down = mag == 0u || (h & 0x8000u);                    // R <= 0: the neighbour below has the larger magnitude
        const uint32_t other = down ? mag + 1u : mag - 1u;
        float fm, fo;
        if constexpr (std::is_same<RT, bf16_t>::value) { fm = bf16_bits_to_float(mag); fo = bf16_bits_to_float(other); }
        else { fm = f16_bits_to_float(mag); fo = f16_bits_to_float(other); }
        const uint32_t mid = __float_as_uint(0.5f * (fm + fo));
        const uint32_t odd = mag & 1u;
        return __uint_as_float(down ? ((mid - odd) | 0x80000000u) : mid + odd);
    }
}
__device__ __forceinline__ float tome_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));          // (no canonicalising v_max_f32 x, x per operand)
    return d;
}
template <typename RT, int QI, int PJ>
__device__ __forceinline__ void tome_running_max2(const tome_f32x16 (&acc)[PJ][QI], float (&bestv)[QI], int (&bestj)[QI], int jlane, int nb,
                                                  bool inside) {
    static_assert(PJ != 4, "the four-wave kernel has its own running max (tome_running_max2_agpr)");
#pragma unroll
    for (int q = 0; q < QI; ++q) {
        float v[PJ * 16];
#pragma unroll
        for (int p = 0; p < PJ; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) v[p * 16 + e] = acc[p][q][e];
        if (!inside) {
#pragma unroll
            for (int p = 0; p < PJ; ++p)
#pragma unroll
                for (int e = 0; e < 16; ++e) v[p * 16 + e] = jlane + p * 32 + (e & 3) + 8 * (e >> 2) < nb ? v[p * 16 + e] : -INFINITY;
        }
        float m = tome_max3(v[0], v[1], v[2]);
#pragma unroll
        for (int e = 3; e + 1 < PJ * 16; e += 2) m = tome_max3(m, v[e], v[e + 1]);
        m = tome_max3(m, v[PJ * 16 - 1], v[PJ * 16 - 1]);
        const float r = tome_round_as<RT>(m);
        const bool gt = r > bestv[q];
        if (__builtin_amdgcn_ballot_w64(gt) != 0ull) {
            const float theta = tome_tie_floor<RT>(r);
            int bc = 0;
#pragma unroll
            for (int p = PJ - 1; p >= 0; --p)
#pragma unroll
                for (int e = 15; e >= 0; --e) bc = v[p * 16 + e] >= theta ? p * 32 + (e & 3) + 8 * (e >> 2) : bc;
            bestj[q] = gt ? jlane + bc : bestj[q];
            bestv[q] = gt ? r : bestv[q];
        }
    }
}

template <typename T, int B> struct TomeAcc;
template <int N, typename F> __device__ __forceinline__ void tome_static_for(F&& f);
// The four-wave kernel's running max: block (p, q) of the AGPR accumulators is copied to 16 VGPRs and swept, one block at a time.
// Branch-free also in a partial tile -- a candidate past nb scores -inf, which is never STRICTLY above the running maximum --, the
// inside / partial decision a scalar branch per 32-candidate block.
template <typename T, int QI, int PJ, typename RoundFn>
__device__ __forceinline__ void tome_running_max_agpr(float (&bestv)[QI], int (&bestj)[QI], int jlane, int jwave, int nb, RoundFn rnd) {
    tome_static_for<QI>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        float bv = bestv[q];
        int bc = -1;
        tome_static_for<PJ>([&](auto P) {
            constexpr int p = decltype(P)::value;
            float o[16];
            TomeAcc<T, p * QI + q>::read(o);
            if (jwave + p * 32 + 32 <= nb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = rnd(o[e]);
                    const bool gt = v > bv;
                    bc = gt ? p * 32 + (e & 3) + 8 * (e >> 2) : bc;
                    bv = gt ? v : bv;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int off = p * 32 + (e & 3) + 8 * (e >> 2);
                    float v = rnd(o[e]);
                    v = jlane + off < nb ? v : -INFINITY;
                    const bool gt = v > bv;
                    bc = gt ? off : bc;
                    bv = gt ? v : bv;
                }
            }
        });
        bestj[q] = bc >= 0 ? jlane + bc : bestj[q];
        bestv[q] = bv;
    });
}

// The same two passes over the AGPR accumulators of the four-wave kernel: pass 1 copies one block of 16 at a time and folds it into the
// row maximum, pass 2 (only when some lane improves) copies the blocks again, last block first.
template <typename T, typename RT, int QI, int PJ>
__device__ __forceinline__ void tome_running_max2_agpr(float (&bestv)[QI], int (&bestj)[QI], int jlane, int jwave, int nb) {
    tome_static_for<QI>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        float m = -INFINITY;
        tome_static_for<PJ>([&](auto P) {
            constexpr int p = decltype(P)::value;
            float o[16];
            TomeAcc<T, p * QI + q>::read(o);
            if (jwave + p * 32 + 32 > nb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = jlane + p * 32 + (e & 3) + 8 * (e >> 2) < nb ? o[e] : -INFINITY;
            }
#pragma unroll
            for (int e = 0; e < 16; e += 2) m = tome_max3(m, o[e], o[e + 1]);
        });
        const float r = tome_round_as<RT>(m);
        const bool gt = r > bestv[q];
        if (__builtin_amdgcn_ballot_w64(gt) != 0ull) {
            const float theta = tome_tie_floor<RT>(r);
            int bc = 0;
            tome_static_for<PJ>([&](auto P) {
                constexpr int p = PJ - 1 - decltype(P)::value;
                float o[16];
                TomeAcc<T, p * QI + q>::read(o);
                if (jwave + p * 32 + 32 > nb) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[e] = jlane + p * 32 + (e & 3) + 8 * (e >> 2) < nb ? o[e] : -INFINITY;
                }
#pragma unroll
                for (int e = 15; e >= 0; --e) bc = o[e] >= theta ? p * 32 + (e & 3) + 8 * (e >> 2) : bc;
            });
            bestj[q] = gt ? jlane + bc : bestj[q];
            bestv[q] = gt ? r : bestv[q];
        }
    });
}

// Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  Logical ids are handed out so
// that one XCD gets a CONTIGUOUS range: the workgroups of one a-tile (its jsplit j-parts, which re-read the same a rows) and of
// neighbouring a-tiles (which stream the same b rows at the same time) then share an L2 instead of each pulling its operands
// from the Infinity Cache.  Bijective for any grid size.
__device__ __forceinline__ int tome_xcd_logical_id() {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__global__ void __launch_bounds__(256, 2) k_tome_match(const float* __restrict__ ahat, const float* __restrict__ bhat,
                                                        int na, int nb, int D, int jsplit,
                                                        unsigned long long* __restrict__ best /*[na]*/) {
    __shared__ float As[TM_K][TM_LD];     // As[k][i]
    __shared__ float Bs[TM_K][TM_LD];     // Bs[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;          // wave tile: 64 (j) x 64 (i)
    const int lid = tome_xcd_logical_id();
    const int itile = lid / jsplit, jpart = lid % jsplit;
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
        tome_running_max<2>(acc, bestv, bestj, j0 + wj * 64 + 4 * lhalf, nb, j0 + TM_J <= nb, [](float v) { return v; });
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
// VEC = 8: one head, D % 8 == 0, 16-byte aligned rows (the production shape): one 16-byte load per lane and chunk, the row stays
// in registers between the two passes (D <= 4096).  VEC = 1: any shape.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_tome_normalize16(const void* __restrict__ x, int n, int C, int n_head, int D, int Dp,
                                                          uint16_t* __restrict__ ahat, uint16_t* __restrict__ bhat) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int row = blockIdx.x * nwave + wave; row < n; row += gridDim.x * nwave) {
        uint16_t* out = ((row & 1) ? bhat : ahat) + (int64_t)(row >> 1) * Dp;
        if constexpr (VEC == 8) {
            constexpr int MAXCH = 8;                     // chunks of 512 elements: D <= 4096
            Pack<T, 8> v[MAXCH];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < MAXCH; ++i) {
                const int d = (i * 64 + lane) * 8;
                if (d < D) {
                    v[i] = load_pack<T, 8>(x, (int64_t)row * C + d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float m = v[i].get(e); ss = fmaf(m, m, ss); }
                }
            }
            ss = wave_sum(ss);
            const float nrm = tome_round<T>(sqrtf(ss));
#pragma unroll
            for (int i = 0; i < MAXCH; ++i) {
                const int d = (i * 64 + lane) * 8;
                if (d < D) {
                    Pack<T, 8> o;
                    const Pack<T, 8> in = v[i];
                    pack_fill<T, 8>(o, [&](int e) { return in.get(e) / nrm; });
                    store_pack<T, 8>(out, d, o);
                }
            }
        } else {
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
        }
        for (int d = D + lane; d < Dp; d += 64) out[d] = 0;
    }
}

typedef __bf16 tome_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tome_f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned int tome_u32x4 __attribute__((ext_vector_type(4)));      // 16-byte staging unit (a native vector: stays in registers)
template <typename T> struct TomeMfma;
template <> struct TomeMfma<bf16_t> {
    typedef tome_bf16x8 vec;
    static __device__ __forceinline__ f32x16_t run(vec a, vec b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct TomeMfma<f16_t> {
    typedef tome_f16x8 vec;
    static __device__ __forceinline__ f32x16_t run(vec a, vec b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// The four-wave kernel's accumulators (round 6): 16 blocks of 16 registers = the WHOLE AGPR file, named literally in inline assembly.
// hipcc never sees them as values -- with the accumulators as C++ variables (through the MFMA builtin, or tied to the "a" register class)
// it keeps part of them in VGPRs, copies all 256 to VGPRs in front of the running max, spills 0.3-1.1 KB per lane and evicts accumulator
// blocks to scratch inside the MFMA sequence.  Every statement lists all AGPRs as clobbers (which also makes the kernel descriptor
// allocate them), and tome.hip is compiled with -amdgpu-spill-vgpr-to-agpr=0, so the compiler holds nothing in an AGPR across them
// (tests/test_kernel_resources.py audits the code object: no v_accvgpr_* outside these statements, no scratch).
// hipcc does not see inside the strings: `s_nop 1` covers a compiler VALU write of an operand right in front of an MFMA; a chain of MFMAs
// on one accumulator needs no wait states; tome_mfma_drain() precedes the first v_accvgpr_read after the last MFMA (8-pass XDL: 12+ states).
#define TOME_AGPR_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <typename T> struct TomeMfmaName;
template <> struct TomeMfmaName<bf16_t> { typedef tome_bf16x8 vec; };
template <> struct TomeMfmaName<f16_t> { typedef tome_f16x8 vec; };
template <typename T, int B> struct TomeAcc {          // accumulator block B = a[16 B : 16 B + 15]
    typedef typename TomeMfmaName<T>::vec vec;
    static __device__ __forceinline__ void mfma(vec a, vec b) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(16 * B), "i"(16 * B + 15) : TOME_AGPR_ALL);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(16 * B), "i"(16 * B + 15) : TOME_AGPR_ALL);
    }
    static __device__ __forceinline__ void zero(vec z) {      // D = 0 * 0 + 0
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" :: "v"(z), "i"(16 * B), "i"(16 * B + 15) : TOME_AGPR_ALL);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c1:%c2], %0, %0, 0" :: "v"(z), "i"(16 * B), "i"(16 * B + 15) : TOME_AGPR_ALL);
    }
    static __device__ __forceinline__ void read(float (&o)[16]) {
        asm volatile("v_accvgpr_read_b32 %0, a%c8\n\tv_accvgpr_read_b32 %1, a%c9\n\tv_accvgpr_read_b32 %2, a%c10\n\tv_accvgpr_read_b32 %3, a%c11\n\t"
                     "v_accvgpr_read_b32 %4, a%c12\n\tv_accvgpr_read_b32 %5, a%c13\n\tv_accvgpr_read_b32 %6, a%c14\n\tv_accvgpr_read_b32 %7, a%c15"
                     : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7])
                     : "i"(16 * B), "i"(16 * B + 1), "i"(16 * B + 2), "i"(16 * B + 3), "i"(16 * B + 4), "i"(16 * B + 5), "i"(16 * B + 6), "i"(16 * B + 7));
        asm volatile("v_accvgpr_read_b32 %0, a%c8\n\tv_accvgpr_read_b32 %1, a%c9\n\tv_accvgpr_read_b32 %2, a%c10\n\tv_accvgpr_read_b32 %3, a%c11\n\t"
                     "v_accvgpr_read_b32 %4, a%c12\n\tv_accvgpr_read_b32 %5, a%c13\n\tv_accvgpr_read_b32 %6, a%c14\n\tv_accvgpr_read_b32 %7, a%c15"
                     : "=v"(o[8]), "=v"(o[9]), "=v"(o[10]), "=v"(o[11]), "=v"(o[12]), "=v"(o[13]), "=v"(o[14]), "=v"(o[15])
                     : "i"(16 * B + 8), "i"(16 * B + 9), "i"(16 * B + 10), "i"(16 * B + 11), "i"(16 * B + 12), "i"(16 * B + 13), "i"(16 * B + 14), "i"(16 * B + 15));
    }
};
__device__ __forceinline__ void tome_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }
template <int N, typename F> __device__ __forceinline__ void tome_static_for(F&& f) {
    if constexpr (N > 0) { tome_static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

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
    const int lid = tome_xcd_logical_id();
    const int itile = lid / jsplit, jpart = lid % jsplit;
    const int i0 = itile * TM_I;
    const int jtiles = (nb + TM_J - 1) / TM_J;
    const int jt_lo = (int)((long long)jtiles * jpart / jsplit), jt_hi = (int)((long long)jtiles * (jpart + 1) / jsplit);
    // staging map: 4 threads per row (4 x 16 bytes = 32 k), 64 rows per pass, 2 passes per 128-row tile
    const int srow = tid >> 2, sch = (tid & 3) * 8;
    float bestv[2] = {-INFINITY, -INFINITY};
    int bestj[2] = {0x7fffffff, 0x7fffffff};
    const int lcol = lane & 31, lhalf = lane >> 5;
    tome_u32x4 ra[2], rb[2];
    auto fetch = [&](int j0, int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ri = min(i0 + p * 64 + srow, na - 1), rj = min(j0 + p * 64 + srow, nb - 1);
            ra[p] = *reinterpret_cast<const tome_u32x4*>(ahat + (int64_t)ri * D + k0 + sch);
            rb[p] = *reinterpret_cast<const tome_u32x4*>(bhat + (int64_t)rj * D + k0 + sch);
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
                *reinterpret_cast<tome_u32x4*>(As + (p * 64 + srow) * TM16_LD + sch) = ra[p];
                *reinterpret_cast<tome_u32x4*>(Bs + (p * 64 + srow) * TM16_LD + sch) = rb[p];
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
        tome_running_max<2>(acc, bestv, bestj, j0 + wj * 64 + 4 * lhalf, nb, j0 + TM_J <= nb, [](float v) { return tome_round<T>(v); });
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = i0 + wi * 64 + q * 32 + lcol;
        if (i < na && bestj[q] != 0x7fffffff) atomicMax(best + i, pack_score(bestv[q], bestj[q]));
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32 inputs on the 16-bit matrix pipe (16x the rate of v_mfma_f32_32x32x2_f32): every unit-row element v is written as
//     4096 * v = h + l + e,   h = fp16(4096 v),  l = fp16(4096 v - h)     (|v| <= 1: no overflow; the scale keeps l normal)
// 4096 v - h is exact in fp32 and has at most 13 significant bits below h's last place, of which l keeps 11: |e| <= 2^-23 |4096 v|
// (zero for most elements).  The score is accumulated in fp32 from the four products  l.l + l.h + h.l + h.h  (every
// fp16 x fp16 product is exact in fp32) and multiplied by 2^-24 at the end.
// Error of one score against the real dot product of the fp32 rows: <= 2 * 2^-23 * sum|a_k b_k| <= 2.4e-7 from e
// (Cauchy-Schwarz, unit rows; three terms drop l.l: + 2^-22, 4.8e-7 in all) plus the fp32 accumulation of the partial sums --
// the same order as any fp32 FMA chain over k = 1024.  Measured on the 128-frame test clip against a float64 product: max
// |error| 9.1e-7 with four AND with three terms, 1.37e-6 for the fp32-input MFMA kernel above, no argmax change in 12 544
// rows.  The reference's own scores are only defined up to that accumulation order (cuBLAS / MKL sgemm).
// ---------------------------------------------------------------------------------------------------
constexpr float kSplitScale = 4096.f;
constexpr float kSplitUnscale = 1.f / (4096.f * 4096.f);

// VEC = 4: one head, D % 4 == 0, 16-byte aligned rows (the production shape): float4 loads, 8-byte plane stores; the second
// pass re-reads the row from the cache.  VEC = 1: any shape.
template <int VEC>
__global__ void __launch_bounds__(256) k_tome_normalize_split(const float* __restrict__ x, int n, int C, int n_head, int D, int Dp,
                                                              uint16_t* __restrict__ aplanes /*[2][na][Dp]*/, int na,
                                                              uint16_t* __restrict__ bplanes /*[2][nb][Dp]*/, int nb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int row = blockIdx.x * nwave + wave; row < n; row += gridDim.x * nwave) {
        const float* xr = x + (int64_t)row * C;
        const bool odd = row & 1;
        uint16_t* hi = (odd ? bplanes : aplanes) + (int64_t)(row >> 1) * Dp;
        uint16_t* lo = hi + (int64_t)(odd ? nb : na) * Dp;
        auto split = [&](float m, float nrm, uint16_t& h_out, uint16_t& l_out) {
            const float v = (m / nrm) * kSplitScale;                  // the reference's unit-row element, times 2^12 (exact)
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);              // v - h is exact in fp32
            h_out = __builtin_bit_cast(uint16_t, h);
            l_out = __builtin_bit_cast(uint16_t, l);
        };
        if constexpr (VEC == 4) {
            // per-lane partial sums over d = 4*lane + e + 256*i in increasing i, e: the order of the scalar loop below is
            // d = lane + 64*i; both feed the same wave_sum tree, the two orders differ only in which lane adds which term
            float ss = 0.f;
            for (int d = lane * 4; d < D; d += 256) {
                const float4 m = *reinterpret_cast<const float4*>(xr + d);
                ss = fmaf(m.x, m.x, ss); ss = fmaf(m.y, m.y, ss); ss = fmaf(m.z, m.z, ss); ss = fmaf(m.w, m.w, ss);
            }
            ss = wave_sum(ss);
            const float nrm = sqrtf(ss);
            for (int d = lane * 4; d < D; d += 256) {
                const float4 m = *reinterpret_cast<const float4*>(xr + d);
                uint16_t h[4], l[4];
                split(m.x, nrm, h[0], l[0]); split(m.y, nrm, h[1], l[1]); split(m.z, nrm, h[2], l[2]); split(m.w, nrm, h[3], l[3]);
                *reinterpret_cast<uint2*>(hi + d) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(lo + d) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
            }
        } else {
            auto metric = [&](int d) {
                if (n_head == 1) return xr[d];
                float s = 0.f;
                for (int h = 0; h < n_head; ++h) s += xr[h * D + d];
                return s / (float)n_head;
            };
            float ss = 0.f;
            for (int d = lane; d < D; d += 64) { const float m = metric(d); ss = fmaf(m, m, ss); }
            ss = wave_sum(ss);
            const float nrm = sqrtf(ss);
            for (int d = lane; d < D; d += 64) split(metric(d), nrm, hi[d], lo[d]);
        }
        for (int d = D + lane; d < Dp; d += 64) { hi[d] = 0; lo[d] = 0; }
    }
}

// KS = k per LDS stage, QI = 32-row a-subtiles per wave, WJ = waves along j.  Workgroup tile: 64*WJ b-rows x 64*QI a-rows,
// 2*WJ waves of 64 (j) x 32*QI (i).  TERMS = 4 (l.l + l.h + h.l + h.h) or 3 (without l.l).
template <int KS, int QI, int WJ, int TERMS>
__global__ void __launch_bounds__(128 * WJ, 2) k_tome_match_split(const uint16_t* __restrict__ ap, const uint16_t* __restrict__ bp,
                                                                   int na, int nb, int D, int jsplit,
                                                                   unsigned long long* __restrict__ best /*[na]*/) {
    constexpr int NT = 128 * WJ;
    constexpr int LD = KS + 8;                 // LDS row: KS fp16 + 16 bytes (ds_read_b128 groups hit distinct 4-bank groups)
    constexpr int TI = 64 * QI, TJ = 64 * WJ;
    constexpr int CPR = KS / 8;                // 16-byte chunks per row and stage
    constexpr int NA = 2 * TI * CPR / NT, NB = 2 * TJ * CPR / NT;
    extern __shared__ __attribute__((aligned(16))) uint16_t tm_smem[];
    uint16_t* As = tm_smem;                    // [plane][TI][LD]
    uint16_t* Bs = tm_smem + 2 * TI * LD;      // [plane][TJ][LD]
    typedef tome_f16x8 vec;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int lid = tome_xcd_logical_id();
    const int itile = lid / jsplit, jpart = lid % jsplit;
    const int i0 = itile * TI;
    const int jtiles = (nb + TJ - 1) / TJ;
    const int jt_lo = (int)((long long)jtiles * jpart / jsplit), jt_hi = (int)((long long)jtiles * (jpart + 1) / jsplit);
    float bestv[QI];
    int bestj[QI];
#pragma unroll
    for (int q = 0; q < QI; ++q) { bestv[q] = -INFINITY; bestj[q] = 0x7fffffff; }
    const int lcol = lane & 31, lhalf = lane >> 5;
    tome_u32x4 ra[NA], rb[NB];
    auto fetch = [&](int j0, int k0) {
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const int id = c * NT + tid, plane = id / (TI * CPR), rem = id % (TI * CPR), row = rem / CPR, ch = rem % CPR;
            const int ri = min(i0 + row, na - 1);
            ra[c] = *reinterpret_cast<const tome_u32x4*>(ap + ((int64_t)plane * na + ri) * D + k0 + ch * 8);
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int id = c * NT + tid, plane = id / (TJ * CPR), rem = id % (TJ * CPR), row = rem / CPR, ch = rem % CPR;
            const int rj = min(j0 + row, nb - 1);
            rb[c] = *reinterpret_cast<const tome_u32x4*>(bp + ((int64_t)plane * nb + rj) * D + k0 + ch * 8);
        }
    };
    if (jt_lo < jt_hi) fetch(jt_lo * TJ, 0);
    for (int jt = jt_lo; jt < jt_hi; ++jt) {
        const int j0 = jt * TJ;
        f32x16_t acc[2][QI];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < QI; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][q][e] = 0.f;
        for (int k0 = 0; k0 < D; k0 += KS) {
            __syncthreads();          // previous stage fully consumed
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int id = c * NT + tid, plane = id / (TI * CPR), rem = id % (TI * CPR), row = rem / CPR, ch = rem % CPR;
                *reinterpret_cast<tome_u32x4*>(As + (plane * TI + row) * LD + ch * 8) = ra[c];
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const int id = c * NT + tid, plane = id / (TJ * CPR), rem = id % (TJ * CPR), row = rem / CPR, ch = rem % CPR;
                *reinterpret_cast<tome_u32x4*>(Bs + (plane * TJ + row) * LD + ch * 8) = rb[c];
            }
            __syncthreads();
            if (k0 + KS < D) fetch(j0, k0 + KS);
            else if (jt + 1 < jt_hi) fetch(j0 + TJ, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ks += 16) {
                vec fb[2][2], fa[QI][2];      // [subtile][plane: 0 = h, 1 = l]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        fb[p][pl] = *reinterpret_cast<const vec*>(Bs + (pl * TJ + wj * 64 + p * 32 + lcol) * LD + ks + lhalf * 8);
#pragma unroll
                    for (int q = 0; q < QI; ++q)
                        fa[q][pl] = *reinterpret_cast<const vec*>(As + (pl * TI + wi * (32 * QI) + q * 32 + lcol) * LD + ks + lhalf * 8);
                }
                // terms outermost: consecutive MFMAs write different accumulators
#pragma unroll
                for (int term = 4 - TERMS; term < 4; ++term) {
                    const int pb = term == 0 || term == 1, pa = term == 0 || term == 2;    // l.l, l.h, h.l, h.h
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int q = 0; q < QI; ++q)
                            acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[p][pb], fa[q][pa], acc[p][q], 0, 0, 0);
                }
            }
        }
        // running max over this tile (on the scaled scores: the factor 2^-24 is applied once, at the end)
        tome_running_max<QI>(acc, bestv, bestj, j0 + wj * 64 + 4 * lhalf, nb, j0 + TJ <= nb, [](float v) { return v; });
    }
#pragma unroll
    for (int q = 0; q < QI; ++q) {
        const int i = i0 + wi * (32 * QI) + q * 32 + lcol;
        if (i < na && bestj[q] != 0x7fffffff) atomicMax(best + i, pack_score(bestv[q] * kSplitUnscale, bestj[q]));
    }
}

// The match on a 256 x 256 workgroup tile (8 waves of 64 (j) x 128 (i), one workgroup per CU) with the operand tiles copied
// global -> LDS by the DMA path (global_load_lds_dwordx4: no staging registers, no ds_write pass) into two LDS buffers.
// A stage is 64 KB (2 x 64 KB buffers of the 160 KB LDS): 32 k of both planes of both matrices for the fp16 split of fp32
// inputs (NP = 2), 64 k of both matrices for bf16 / fp16 inputs (NP = 1).  LDS reads per MFMA are 2/3 of the 128 x 128 kernels'.
// An LDS-DMA instruction writes wave-linear (base + lane * 16 bytes), so rows are unpadded (64 / 128 bytes) and the bank spread
// comes from a chunk swizzle applied to the SOURCE address: 16-byte chunk c of tile row r lives at chunk position
// c ^ ((r / rows-per-256-bytes) & (chunks-per-row - 1)), which sends the 16 rows of a ds_read_b128 lane group to 16 different
// 4-bank groups (SQ_LDS_BANK_CONFLICT = 0).
// Stage pipeline: every k16 step's operand fragments are read from LDS while the MFMAs of the step before run (two fragment
// register sets), ACROSS the stage barrier -- the barrier of stage s+1 sits in front of the last step's MFMAs of stage s, so the
// first fragments of stage s+1 are read under those MFMAs, and the DMA of stage s+2 (into the buffer everyone has just finished
// reading) is issued there as well, two pieces after every group of 8 MFMAs.
// NP = 2: fp32 inputs as two fp16 planes, TERMS = 4 / 3 products per k (T = f16_t).  NP = 1: 16-bit inputs of type T, one
// product, scores rounded to T before the comparison (the reference's score tensor has the input dtype).
typedef const __attribute__((address_space(1))) void* tome_gptr;
typedef __attribute__((address_space(3))) void* tome_lptr;
constexpr int TG_T = 256;                       // tile side
constexpr int TG_BUF = 65536;                   // bytes of one stage: [matrix A, B][plane][256 rows][KS * 2 bytes]

// Round 3: (a) the running max of a tile product that lies inside [0, nb) is branch-free -- a lane meets its candidates in
// ascending j (p, then e), so "first maximum" is a strict `>` and costs a compare and two selects per score; written as
// `if (j < nb && (v > best || (v == best && j < bestj)))` the compiler produced ~30 instructions and an exec-mask branch per score,
// 3 800 instructions per wave and tile product (23 % of a four-term product's MFMA time, about as long as a one-term product's);
// (b) one-plane kernels request step n+1's fragments after the first four MFMAs of step n: the compiler waits with lgkmcnt(0) in
// front of a step's first MFMA, so reads issued before it were waited for at once (8 MFMAs per step cannot hide that; the 32 of the
// two-plane kernels can: no difference there).  Results are bit-identical.
// ABL (dev builds, STTM_TOME_ABL): 1 = no DMA after the prologue (MFMA side alone), 2 = no MFMAs (DMA + barriers + fragment reads alone),
// 5 = MFMAs alone (no fragment reads, barriers or DMA in the loop), 6 = MFMAs + fragment reads (no barriers, no DMA): outputs invalid
// in all four; 3 = the round-2 form (general running max everywhere, reads in front), 4 = reads after the FIRST MFMA.
// PJ = 32-row b-subtiles per wave (round 6): 2 = eight waves of 64 (j) x 128 (i), two per SIMD; 4 = FOUR waves of 128 x 128, one per SIMD with
// the whole 512-register budget -- its 256 accumulators in the AGPR half, 16 MFMAs per k16 step and plane product between two fragment waits,
// a quarter fewer LDS fragment bytes per MFMA, no second wave contending for the SIMD's issue slots.
template <int NP, int TERMS, typename T, int ABL = 0, int PJ = 2>
__global__ void __launch_bounds__(128 * (8 / PJ), PJ == 4 ? 1 : 2) k_tome_match_glds(const uint16_t* __restrict__ ap, const uint16_t* __restrict__ bp,
                                                             int na, int nb, int D, int jsplit,
                                                             unsigned long long* __restrict__ best /*[na]*/) {
    static_assert((NP == 2 && (TERMS == 3 || TERMS == 4)) || (NP == 1 && TERMS == 1), "plane / term combination");
    extern __shared__ __attribute__((aligned(1024))) char tg_smem[];
    typedef typename TomeMfma<T>::vec vec;
    constexpr int QI = 4;
    constexpr int KS = 64 / NP;                  // k per stage
    constexpr int NSTEP = KS / 16;               // k16 steps per stage (even)
    constexpr int RB = KS * 2;                   // bytes per tile row
    constexpr int CPR = RB / 16;                 // 16-byte chunks per row
    constexpr int RPP = 1024 / RB;               // rows per 1 KB piece
    constexpr int PPM = TG_T / RPP;              // pieces per plane of one matrix tile
    constexpr int PLANE = TG_T * RB;             // bytes of one plane of one matrix tile
    constexpr int R256 = 256 / RB;               // rows per 256 bytes of LDS (the bank period)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWAVE = 2 * (8 / PJ);          // 2 along i x (4 or 2) along j
    constexpr int NPIECE = 64 / NWAVE;           // 1 KB DMA pieces per wave and stage
    const int wi = wave & 1, wj = wave >> 1;
    const int lid = tome_xcd_logical_id();
    // A workgroup owns a contiguous range [p_lo, p_hi) of tile products in row-major order (p = itile * jtiles + jtile).
    // jsplit > 0: the j tiles of ONE a-tile in jsplit parts (grid = itiles * jsplit).  jsplit == 0: all products spread evenly over
    // the grid (one workgroup per CU), a range may cross into the next a-tile(s) -- 69 x 69 tiles (T = 180) on 256 CUs are 23 products
    // per workgroup on 207 workgroups the first way and 18 or 19 on 256 the second.  Either way a lane meets the candidates of an
    // a-row in ascending j; rows are published (packed atomicMax) and the running max reset whenever the a-tile changes.
    const int jtiles = (nb + TG_T - 1) / TG_T;
    int p_lo, p_hi;
    if (jsplit > 0) {
        const int itile = lid / jsplit, jpart = lid % jsplit;
        p_lo = itile * jtiles + (int)((long long)jtiles * jpart / jsplit);
        p_hi = itile * jtiles + (int)((long long)jtiles * (jpart + 1) / jsplit);
    } else {
        const long long total = (long long)((na + TG_T - 1) / TG_T) * jtiles;
        p_lo = (int)(total * lid / gridDim.x);
        p_hi = (int)(total * (lid + 1) / gridDim.x);
    }
    const int lcol = lane & 31, lhalf = lane >> 5;
    const int NK = D / KS;                        // stages per tile product
    const int S = (p_hi - p_lo) * NK;             // stages of this workgroup
    if (S <= 0) return;

    const int prow = lane / CPR;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ap), 0, NP * na * D * 2, 0x00020000);      // (< 2 GiB: tome_plan)
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bp), 0, NP * nb * D * 2, 0x00020000);
    const uint32_t dma_voff = (uint32_t)((prow * D + ((lane % CPR) ^ ((4 * wave + prow / R256) & (CPR - 1))) * 8) * 2);
    // Round 6: buffer-addressed DMA with a RUNNING stage state.  The piece's first row and k0 are a SCALAR byte offset, every lane
    // adds ONE offset that is the same for all of the wave's pieces (the source-chunk swizzle depends on the piece only through its parity,
    // which is the wave's), rows past the matrix are cut off by the descriptor's range check (they read zeros, or -- first plane of two --
    // the other plane's rows: scores of rows >= na are never published, candidates >= nb never compared).  No 64-bit per-lane address
    // arithmetic and no per-piece VGPRs; the (a-tile, j-tile, k-stage) of a stage is carried from stage to stage instead of being divided
    // out of the stage number per piece (st / NK, prod / jtiles with run-time divisors: ~45 scalar instructions a piece, 700 a stage --
    // with one wave per SIMD nothing else issues meanwhile).
    struct Feed { int it, jt, k; };
    auto feed_next = [&](Feed& f) {
        if (++f.k == NK) { f.k = 0; if (++f.jt == jtiles) { f.jt = 0; ++f.it; } }
    };
    auto issue_piece4 = [&](int c, const Feed& f, int buf, bool prologue = false) {
        if ((ABL == 1 || ABL == 5 || ABL == 6) && !prologue) return;
        constexpr int HALF = NPIECE / 2;
        const int cc = c % HALF;
        const bool is_a = c < HALF;
        const int g = wave + NWAVE * cc + (is_a ? 0 : 32);
        const int plane = ((NWAVE * cc) / PPM) % NP, rr = (NWAVE * cc) % PPM;          // (wave < NWAVE <= PPM: the wave does not change them)
        const int row0 = plane * (is_a ? na : nb) + (is_a ? f.it : f.jt) * TG_T + (wave + rr) * RPP;
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((row0 * D + f.k * KS) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rs_a : rs_b, (tome_lptr)(tg_smem + buf * TG_BUF + g * 1024), 16, dma_voff, soff, 0, 0);
    };
    const int sw = (lcol / R256) & (CPR - 1);
    const int a_off = (wi * 128 + lcol) * RB, b_off = NP * PLANE + (wj * 32 * PJ + lcol) * RB;
    struct Frag { vec b[PJ][NP], a[QI][NP]; };
    auto read_frag = [&](Frag& f, int st, int step) {
        if (ABL == 5 && st + step > 0) return;                 // MFMAs alone: the first fragments for ever
        const char* base = tg_smem + (st & 1) * TG_BUF;
        const int coff = ((step * 2 + lhalf) ^ sw) * 16;
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            const int pl = NP - 1 - pi;         // the l planes first: the first products need them
#pragma unroll
            for (int p = 0; p < PJ; ++p) f.b[p][pl] = *reinterpret_cast<const vec*>(base + b_off + pl * PLANE + p * 32 * RB + coff);
#pragma unroll
            for (int q = 0; q < QI; ++q) f.a[q][pl] = *reinterpret_cast<const vec*>(base + a_off + pl * PLANE + q * 32 * RB + coff);
        }
    };

    float bestv[QI];
    int bestj[QI];
#pragma unroll
    for (int q = 0; q < QI; ++q) { bestv[q] = -INFINITY; bestj[q] = 0x7fffffff; }
    // PJ == 2: accumulators as values (hipcc's registers).  PJ == 4: the literal AGPR blocks of TomeAcc, cleared by an MFMA of zero operands
    // with C = 0 (16 of the 1024+ MFMAs of a tile product; taking C = 0 in the product's first MFMA needs a branch between two asm forms
    // in the k-loop).
    constexpr int NACC = PJ == 4 ? 1 : PJ;
    f32x16_t acc[NACC][QI];
    auto clear_acc = [&]() {
        if constexpr (PJ == 4) {
            vec z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = 0;
            tome_static_for<PJ * QI>([&](auto B) { TomeAcc<T, decltype(B)::value>::zero(z); });
        } else {
#pragma unroll
            for (int p = 0; p < PJ; ++p)
#pragma unroll
                for (int q = 0; q < QI; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[p][q][e] = 0.f;
        }
    };
    clear_acc();

    // prologue: stages 0 and 1 in flight, stage 0 landed, its first fragments read
    Feed f1 = {p_lo / jtiles, p_lo % jtiles, 0}, f2 = f1;          // the stages st + 1 and st + 2 of the loop below
#pragma unroll
    for (int c = 0; c < NPIECE; ++c) issue_piece4(c, f1, 0, true);
    feed_next(f1);
    if (S > 1) {
#pragma unroll
        for (int c = 0; c < NPIECE; ++c) issue_piece4(c, f1, 1, true);
    }
    f2 = f1;
    feed_next(f2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frag fr[2];
    read_frag(fr[0], 0, 0);
    if (ABL == 5) fr[1] = fr[0];

    for (int st = 0; st < S; ++st) {
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            Frag& cur = fr[step & 1];
            Frag& nxt = fr[(step + 1) & 1];
            const bool last = step == NSTEP - 1;
            __builtin_amdgcn_sched_barrier(0);      // the MFMAs of the step before stay in front of this step's barrier / reads
            // (PJ == 4: hipcc counts the fragment waits in front of the asm MFMAs one by one -- lgkmcnt(11) .. (8) with the next step's eight
            // reads already behind them --, so the reads can go first and get the whole step's 16 MFMAs to land)
            constexpr int late_n = (NP == 1 && ABL != 3 && PJ != 4) ? (ABL == 4 ? 1 : QI) : 0;   // MFMAs in front of the next step's fragment reads
            constexpr bool late = late_n > 0;
            if (!last) {
                if (!late) read_frag(nxt, st, step + 1);
            } else {
                // every wave has read all of stage st (the reads of this step were issued a step ago; __syncthreads waits for
                // them); stage st+1 has landed once every wave's own pieces have
                if (ABL != 5 && ABL != 6) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                if (st + 1 < S) read_frag(nxt, st + 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // the reads stay in front of the MFMAs they overlap with
            const bool feed = last && st + 2 < S;   // this step also issues the DMA of stage st+2 (into the buffer just vacated)
#pragma unroll
            for (int term = 4 - TERMS; term < 4; ++term) {
                // l.l, l.h, h.l, h.h (small terms first); one plane: the only term is h.h
                const int pb = NP == 2 && (term == 0 || term == 1), pa = NP == 2 && (term == 0 || term == 2);
                if constexpr (PJ == 4) {
                    tome_static_for<PJ * QI>([&](auto B) {
                        constexpr int b = decltype(B)::value, p = b / QI, q = b % QI;
                        TomeAcc<T, b>::mfma(cur.b[p][pb], cur.a[q][pa]);
                        // One wave per SIMD: nobody else covers a burst of DMA issues (~60 cycles each against an MFMA's 32), so the pieces
                        // go out one at a time behind MFMAs.  The buffer of stage st is free from the barrier in front of its last step
                        // until the first reads of stage st+2, one stage later: the pieces of stage st+2 are spread over the first
                        // NSTEP-1 steps of that window -- this stage's last step and the next stage's steps 0 .. NSTEP-3 --, the window's
                        // last step is left for the last piece to land.
                        {
                            constexpr int WSTEPS = NSTEP > 2 ? 2 : 1;                         // window steps that issue
                            constexpr int EVERY = WSTEPS * PJ * QI * TERMS / NPIECE;          // MFMAs per piece (3)
                            const int wpos = last ? 0 : step + 1;
                            const int tgt = last ? st + 2 : st + 1;
                            const int m = (wpos * TERMS + term - (4 - TERMS)) * PJ * QI + b;  // MFMA number inside the window (a constant once unrolled)
                            if (wpos < WSTEPS && tgt >= 2 && tgt < S && m % EVERY == 0 && m / EVERY < NPIECE) issue_piece4(m / EVERY, last ? f2 : f1, tgt & 1);
                        }
                        if (late_n == QI && !last && term == 4 - TERMS && b == QI - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            read_frag(nxt, st, step + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                } else
#pragma unroll
                for (int p = 0; p < PJ; ++p) {
#pragma unroll
                    for (int q = 0; q < QI; ++q) {
                        if (ABL != 2) acc[p][q] = TomeMfma<T>::run(cur.b[p][pb], cur.a[q][pa], acc[p][q]);
                        else asm volatile("" :: "v"(cur.b[p][pb]), "v"(cur.a[q][pa]));      // the fragment reads stay
                        if (late_n == 1 && !last && term == 4 - TERMS && p == 0 && q == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            read_frag(nxt, st, step + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (late_n == QI && !last && term == 4 - TERMS && p == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        read_frag(nxt, st, step + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (last && PJ != 4) {
                    const int gi = term - (4 - TERMS);
                    if (feed) {
#pragma unroll
                        for (int c = 0; c < NPIECE; ++c)
                            if (c * TERMS / NPIECE == gi) issue_piece4(c, f2, st & 1);
                    }
                }
            }
        }
        f1 = f2;
        feed_next(f2);
        if ((st + 1) % NK == 0) {
            // end of a tile product: running max (split: on the scaled scores, the factor 2^-24 is applied once at the end)
            const int prod = p_lo + st / NK, it = prod / jtiles, jt = prod - it * jtiles;
            const int j0 = jt * TG_T;
            typedef typename std::conditional<NP == 1, T, float>::type RT;           // scores are rounded to the input dtype; split planes: fp32
            if constexpr (ABL == 8) {                                                 // (development builds: the one-pass sweep, for the A/B)
                auto rnd_fn = [](float v) { return tome_round_as<RT>(v); };
                if constexpr (PJ == 4) {
                    tome_mfma_drain();
                    tome_running_max_agpr<T, QI, PJ>(bestv, bestj, j0 + wj * 32 * PJ + 4 * lhalf, j0 + wj * 32 * PJ, nb, rnd_fn);
                } else {
                    tome_running_max<QI, decltype(rnd_fn), NACC>(acc, bestv, bestj, j0 + wj * 32 * PJ + 4 * lhalf, nb, j0 + TG_T <= nb, rnd_fn);
                }
            } else if constexpr (PJ == 4) {
                tome_mfma_drain();
                if (ABL != 7) tome_running_max2_agpr<T, RT, QI, PJ>(bestv, bestj, j0 + wj * 32 * PJ + 4 * lhalf, j0 + wj * 32 * PJ, nb);
            } else {
                tome_running_max2<RT, QI, NACC>(acc, bestv, bestj, j0 + wj * 32 * PJ + 4 * lhalf, nb, ABL != 3 && j0 + TG_T <= nb);
            }
            clear_acc();
            if (st + 1 == S || jt + 1 == jtiles) {
                // last product of this a-tile in the range: publish its rows, start over for the next a-tile
#pragma unroll
                for (int q = 0; q < QI; ++q) {
                    const int i = it * TG_T + wi * 128 + q * 32 + lcol;
                    if (i < na && bestj[q] != 0x7fffffff)
                        atomicMax(best + i, pack_score(NP == 2 ? bestv[q] * kSplitUnscale : bestv[q], bestj[q]));
                    bestv[q] = -INFINITY; bestj[q] = 0x7fffffff;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// merge bookkeeping: sources (rank < r) grouped per destination b-token, in rank order
// ---------------------------------------------------------------------------------------------------
__global__ void k_tome_fill(const int* __restrict__ order, const int* __restrict__ node_idx, int r, const int* __restrict__ off,
                            int* __restrict__ cur, int* __restrict__ lists /* rank positions k */) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < r) {
        const int d = node_idx[order[k]];
        lists[off[d] + atomicAdd(cur + d, 1)] = k;
    }
}

// ---------------------------------------------------------------------------------------------------
// Ranking by counting (round 4; round 6: the only ranking path -- it replaced a k_tome_unpack + hipcub::DeviceRadixSort + count + scan
// chain of 8 launches, first for clips of up to 49 152 a-tokens, now for every length: both the ranking and the match are O(na^2) and
// the ranking stays a few per cent of the match).  argsort(node_max, descending) with ties to the smaller index (tome_token_merger.py:37):
//   rank_i = #{ j : key_j > key_i }  on the 64-bit keys  (order-preserving bits of the score << 32) | ~i,
// which are all distinct.  12 544 a-tokens (T = 128) are 1.6e8 compares -- microseconds of VALU time.  Workgroup (ib, js): 256
// a-tokens (one per thread) against the js-th part of the keys, staged through LDS in tiles and read back as broadcasts; partial
// counts go to rank[js][i], and the LAST part of an a-block to arrive (per-block arrival counter) sums them, turns the complete ranks
// into `order`, writes the unpacked node_max / node_idx and counts the sources of every destination (rank < r); the a-block that
// completes last of all scans those counts.
// ---------------------------------------------------------------------------------------------------
constexpr int RK_I = 256;
constexpr int kRankParts = 32;         // most key-range parts per a-block (partial ranks are [parts][na] ints of workspace)

__device__ __forceinline__ unsigned long long tome_rank_key(unsigned long long b, int i) {
    const unsigned ku = b ? (unsigned)(b >> 32) : 0xffc00000u;          // no finite score at all: a NaN row (sorts first)
    return ((unsigned long long)ku << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
}

__global__ void __launch_bounds__(RK_I) k_tome_rank(const unsigned long long* __restrict__ best, int na, int jsplit, int r,
                                                    int* __restrict__ rank, int* __restrict__ arrive, float* __restrict__ node_max,
                                                    int* __restrict__ node_idx, int* __restrict__ order, int* __restrict__ cnt,
                                                    int nb, int* __restrict__ off) {
    __shared__ __attribute__((aligned(16))) unsigned tile[RK_I];
    __shared__ int last_sh, all_sh;
    __shared__ int wsum[RK_I / 64];
    __shared__ int stile[16 * RK_I];            // the closing scan's transpose tile (16 KB)
    const int ib = blockIdx.x / jsplit, js = blockIdx.x - ib * jsplit;
    const int tid = threadIdx.x;
    const int i = ib * RK_I + tid;
    const unsigned long long bi = i < na ? best[i] : 0ull;
    const unsigned long long ki = tome_rank_key(bi, i);
    const int chunk = ((na + jsplit - 1) / jsplit + RK_I - 1) / RK_I * RK_I;
    const int jlo = js * chunk, jhi = min(na, jlo + chunk);
    // Keys are compared as 32-bit score bits; the index tie-break (j < i) is UNIFORM for every 256-key block other than the
    // workgroup's own one (blocks are aligned): blocks before it count `>=`, blocks after it `>`, one compare + one add per key;
    // only the diagonal block takes the three-compare form.  Keys are staged in LDS as 32-bit words and read back four per
    // broadcast.  (First version: 64-bit keys and compares, one LDS broadcast per key: 67 us for 12.5 k keys; j-side in scalar
    // registers through the scalar cache: 127 us.)
    const unsigned kui = (unsigned)(ki >> 32);
    int c = 0;
    for (int j0 = jlo; j0 < jhi; j0 += RK_I) {       // one aligned block of 256 keys per step
        const int jj = j0 + tid;
        tile[tid] = jj < jhi ? (unsigned)(tome_rank_key(best[jj], jj) >> 32) : 0u;      // 0 is below every key
        __syncthreads();
        const int nq = min(RK_I, jhi - j0);
        if (j0 == ib * RK_I) {                       // the diagonal block
            for (int q = 0; q < nq; ++q) {
                const unsigned kj = tile[q];
                c += (kj > kui || (kj == kui && j0 + q < i)) ? 1 : 0;
            }
        } else if (j0 < ib * RK_I) {                 // every j of the block is before every i of the workgroup
#pragma unroll 8
            for (int q = 0; q < RK_I; q += 4) {
                const uint4 k4 = *reinterpret_cast<const uint4*>(tile + q);
                c += (k4.x >= kui) + (k4.y >= kui) + (k4.z >= kui) + (k4.w >= kui);
            }
        } else {
#pragma unroll 8
            for (int q = 0; q < RK_I; q += 4) {
                const uint4 k4 = *reinterpret_cast<const uint4*>(tile + q);
                c += (k4.x > kui) + (k4.y > kui) + (k4.z > kui) + (k4.w > kui);
            }
        }
        __syncthreads();
    }
    // Ordering contract of this hand-off (round-4 advisor note): it is NOT the HIP memory model's release / acquire but the gfx950 form
    // MI355X_MICROARCH.md lists as valid ("handoff-flag: sc1 payload -> asm s_waitcnt vmcnt(0) -> sc1 flag", consumer side sc1 loads): the
    // agent-scope relaxed stores / loads below compile to global_store / global_load ... sc1 (write-through to / read from the device's
    // coherence point, past the CU's L1 and the XCD's L2 copy), the drain makes the payload leave before the arrival atomic, which is a
    // memory-side read-modify-write.  The library is built for gfx950 only (sttm_common.h refuses any other device target);
    // test_tome_ranking_equals_a_stable_descending_argsort compares the order this kernel leaves with torch's stable argsort.
    // Everything that crosses workgroups here is an agent-scope atomic (performed at the device's coherence point) read back with
    // agent-scope loads: every wave drains its own (vmcnt), then ONE relaxed arrival -- no release / acquire fences, which on this
    // multi-XCD part are L2 write-backs + invalidates per workgroup (a __threadfence() here made the kernel 67 - 120 us).
    // (partial counts as plain coalesced write-through stores into rank[js][i], summed by the last part to arrive: as atomics on
    // rank[i] they were 25 memory-side read-modify-writes per a-token, most of the kernel's 33 us)
    if (i < na) __hip_atomic_store(rank + (size_t)js * na + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_sh = __hip_atomic_fetch_add(arrive + ib, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == jsplit - 1;
    __syncthreads();
    if (!last_sh) return;
    if (i < na) {
        int rk = 0;
        for (int q0 = 0; q0 < jsplit; q0 += 8) {           // eight independent loads per round trip
            int pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                pv[u] = q0 + u < jsplit ? __hip_atomic_load(rank + (size_t)(q0 + u) * na + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) rk += pv[u];
        }
        order[rk] = i;
        unsigned u = (unsigned)(bi >> 32);
        u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        const int dst = bi ? (int)(0xffffffffu - (unsigned)(bi & 0xffffffffu)) : 0;
        node_max[i] = bi ? __uint_as_float(u) : __uint_as_float(0x7fc00000u);
        node_idx[i] = dst;
        if (rk < r) atomicAdd(cnt + dst, 1);
    }
    // the a-block that completes LAST of all (slot `iblocks` of the arrival counters) scans the per-destination counts: the old
    // one-workgroup k_tome_scan, without a launch of its own
    const int iblocks = (na + RK_I - 1) / RK_I;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) all_sh = __hip_atomic_fetch_add(arrive + iblocks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == iblocks - 1;
    __syncthreads();
    if (!all_sh) return;
    // chunks of 16 x 256 counts: sixteen independent coalesced loads per thread (ONE round trip per chunk; a thread walking its own
    // contiguous run did one dependent round trip per element -- 2 x 49 of them at T = 128), transposed through LDS so that every
    // thread scans sixteen CONTIGUOUS counts, then the usual wave / workgroup prefix and a running carry
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int PERT = 16, CHUNK = PERT * RK_I;
    int carry = 0;
    for (int base = 0; base < nb; base += CHUNK) {
        int v[PERT];
#pragma unroll
        for (int k = 0; k < PERT; ++k) {
            const int q = base + k * RK_I + tid;
            v[k] = q < nb ? __hip_atomic_load(cnt + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
#pragma unroll
        for (int k = 0; k < PERT; ++k) stile[k * RK_I + tid] = v[k];
        __syncthreads();
        int total = 0;
#pragma unroll
        for (int k = 0; k < PERT; ++k) { v[k] = stile[tid * PERT + k]; total += v[k]; }
        int inc = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int pre = carry + inc - total, all = 0;
        for (int w = 0; w < RK_I / 64; ++w) { if (w < wave) pre += wsum[w]; all += wsum[w]; }
#pragma unroll
        for (int k = 0; k < PERT; ++k) {
            const int q = base + tid * PERT + k;
            if (q < nb) off[q] = pre;
            pre += v[k];
        }
        carry += all;
        __syncthreads();
    }
    if (tid == 0) off[nb] = carry;
}

// one wave per output row.  rows [0, na - r): unmerged a-tokens in rank order; rows [na - r, n - r): b-tokens.
// Every tensor of the reference is rounded to the input dtype T (x * size, the scatter-added sums, the sums of sizes, the
// quotient): for T = float that is the plain fp32 arithmetic of the reference, without contraction.
template <typename T, int VEC> __device__ __forceinline__ void tome_ld_vec(const void* x, int64_t off, float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        v[0] = tome_ld<T>(x, off);
    } else {
        const Pack<T, VEC> p = load_pack<T, VEC>(x, off);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = p.get(e);
    }
}
template <typename T, int VEC> __device__ __forceinline__ void tome_st_vec(void* x, int64_t off, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        tome_st<T>(x, off, v[0]);
    } else {
        Pack<T, VEC> p;
        pack_fill<T, VEC>(p, [&](int e) { return v[e]; });
        store_pack<T, VEC>(x, off, p);
    }
}
// VEC channels per lane and step (VEC * sizeof(T) bytes per access); the per-element arithmetic does not depend on VEC.
// Round 4: 16-byte accesses for every dtype (16-bit rows: 8 channels per lane), up to NCH row chunks of a token in flight per lane
// (C = 1024: the whole row), and a destination's sources looked up ONCE by the lanes in parallel (rank -> token -> size) instead
// of two dependent loads per source and chunk -- the old form moved T = 128 in 28-38 us = 2.3-3 TB/s.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_tome_merge(const void* __restrict__ x, const float* __restrict__ size,
                                                    const int64_t* __restrict__ idx, int n, int C, int na, int nb, int r,
                                                    const int* __restrict__ order, const int* __restrict__ off,
                                                    int* __restrict__ lists, void* __restrict__ x_out,
                                                    float* __restrict__ size_out, int64_t* __restrict__ idx_out) {
#pragma clang fp contract(off)      // x*size is rounded before it is added, like the reference's mul then scatter-add

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int n_out = n - r;
    constexpr int NCH = VEC >= 8 ? 2 : 4;   // chunks of 64 * VEC channels handled together (fp32 / 16-bit C = 1024: the whole row)
    for (int row = blockIdx.x * nwave + wave; row < n_out; row += gridDim.x * nwave) {
        if (row < na - r) {
            const int tok = 2 * order[r + row];                   // unmerged a-token: (x*size)/size
            const float s = size ? size[tok] : 1.f;
            for (int c0 = 0; c0 < C; c0 += NCH * 64 * VEC) {
                float v[NCH][VEC];
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int c = c0 + (u * 64 + lane) * VEC;
                    if (c < C) tome_ld_vec<T, VEC>(x, (int64_t)tok * C + c, v[u]);
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int c = c0 + (u * 64 + lane) * VEC;
                    if (c < C) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) v[u][e] = tome_round<T>(v[u][e] * s) / s;
                        tome_st_vec<T, VEC>(x_out, (int64_t)row * C + c, v[u]);
                    }
                }
            }
            if (lane == 0) { size_out[row] = s; idx_out[row] = idx ? idx[tok] : (int64_t)tok; }
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
        // 16-bit inputs: the sum of a destination's sources is accumulated in fp32 and rounded ONCE to the tensor dtype.  This is
        // a DETERMINISTIC CHOICE, not the reference's only behaviour: ATen's CPU scatter_add takes that route on its
        // expanded-index path (FBGEMM / OpenMP builds: the build that generated tests/golden/tome16_*, recorded in their meta),
        // while its CUDA/HIP kernel and CPU builds without that path round after every atomic add, on the GPU in a
        // nondeterministic order.  For destinations with >= 2 sources the two differ by 1-2 ulp of the dtype, which is the spread
        // of the reference itself from run to run; the parity tests compare such rows with an ulp tolerance, not bit equality.
        // The products x * size are tensors of their own and are rounded individually (both routes agree on that).
        const float sb = size ? size[tok] : 1.f;
        float stot = sb;
        if (cnt <= 64) {
            // lane m holds source m (ascending rank): its token and size, fetched by all lanes at once
            int atok_l = 0;
            float sa_l = 1.f;
            if (lane < cnt) {
                atok_l = 2 * order[lists[o + lane]];
                sa_l = size ? size[atok_l] : 1.f;
            }
            for (int m = 0; m < cnt; ++m) stot = stot + __shfl(sa_l, m, 64);
            stot = tome_round<T>(stot);
            for (int c0 = 0; c0 < C; c0 += NCH * 64 * VEC) {
                float acc[NCH][VEC], xa[NCH][VEC];
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int c = c0 + (u * 64 + lane) * VEC;
                    if (c < C) tome_ld_vec<T, VEC>(x, (int64_t)tok * C + c, acc[u]);
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[u][e] = tome_round<T>(acc[u][e] * sb);      // plain operators: the __fmul_rn/__fadd_rn wrappers fuse once inlined
                for (int m = 0; m < cnt; ++m) {
                    const int atok = __shfl(atok_l, m, 64);
                    const float sa = __shfl(sa_l, m, 64);
#pragma unroll
                    for (int u = 0; u < NCH; ++u) {
                        const int c = c0 + (u * 64 + lane) * VEC;
                        if (c < C) tome_ld_vec<T, VEC>(x, (int64_t)atok * C + c, xa[u]);
                    }
#pragma unroll
                    for (int u = 0; u < NCH; ++u) {
                        const int c = c0 + (u * 64 + lane) * VEC;
                        if (c < C) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) acc[u][e] = acc[u][e] + tome_round<T>(xa[u][e] * sa);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int c = c0 + (u * 64 + lane) * VEC;
                    if (c < C) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[u][e] = tome_round<T>(acc[u][e]) / stot;
                        tome_st_vec<T, VEC>(x_out, (int64_t)row * C + c, acc[u]);
                    }
                }
            }
        } else {
            for (int m = 0; m < cnt; ++m) {
                const int atok = 2 * order[lists[o + m]];
                stot = stot + (size ? size[atok] : 1.f);
            }
            stot = tome_round<T>(stot);
            for (int c = lane * VEC; c < C; c += 64 * VEC) {
                float acc[VEC], xa[VEC];
                tome_ld_vec<T, VEC>(x, (int64_t)tok * C + c, acc);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = tome_round<T>(acc[e] * sb);
                for (int m = 0; m < cnt; ++m) {
                    const int atok = 2 * order[lists[o + m]];
                    const float sa = size ? size[atok] : 1.f;
                    tome_ld_vec<T, VEC>(x, (int64_t)atok * C + c, xa);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = acc[e] + tome_round<T>(xa[e] * sa);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = tome_round<T>(acc[e]) / stot;
                tome_st_vec<T, VEC>(x_out, (int64_t)row * C + c, acc);
            }
        }
        if (lane == 0) { size_out[row] = stot; idx_out[row] = idx ? idx[tok] : (int64_t)tok; }
    }
}

struct TomePlan {
    int na, nb, D, Dp;
    size_t off_ahat, off_bhat, off_best, off_nmax, off_nidx, off_order, off_cnt, off_cur, off_rank, off_arrive, off_off, off_parts, off_lists, total;
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
    p->Dp = (p->D + 63) / 64 * 64;                           // zero-padded to the widest k stage of the match kernels
    // the 256-tile match kernels address a unit-row matrix through ONE buffer descriptor with 32-bit byte offsets (two 2-byte planes per
    // element at most): clips beyond 2 GiB per matrix (fp32 C = 1024: over a million tokens) are refused, not wrapped around
    if ((unsigned long long)p->na * p->Dp * 4ull >= (1ull << 31)) return -2;
    size_t o = 0;
    p->off_ahat = o; o = al(o + (size_t)p->na * p->Dp * 4);
    p->off_bhat = o; o = al(o + (size_t)(p->nb > 0 ? p->nb : 1) * p->Dp * 4);
    p->off_nmax = o; o = al(o + (size_t)p->na * 4);
    p->off_nidx = o; o = al(o + (size_t)p->na * 4);
    p->off_order = o; o = al(o + (size_t)p->na * 4);
    // ONE zeroed region [off_best, off_off): the packed best scores, the per-destination counts and cursors, the partial ranks and
    // the per-block arrival counters of the ranking kernel
    p->off_best = o; o = al(o + (size_t)p->na * 8);
    p->off_cnt = o; o = al(o + (size_t)(p->nb + 1) * 4);
    p->off_cur = o; o = al(o + (size_t)(p->nb + 1) * 4);
    p->off_rank = o; o = al(o + (size_t)p->na * 4);               // (not part of the zeroed region any more: see off_parts)
    p->off_arrive = o; o = al(o + (size_t)((p->na + RK_I - 1) / RK_I + 1) * 4);
    p->off_off = o; o = al(o + (size_t)(p->nb + 2) * 4);
    p->off_lists = o; o = al(o + (size_t)p->na * 4);
    p->off_parts = o; o = al(o + (size_t)kRankParts * p->na * 4);      // partial ranks [parts][na]
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
    if (!x_ || !workspace || !x_out_ || !size_out || !idx_out) return STTM_ERR_ARG;       // (idx == NULL: the identity, first iteration)
    TomePlan p;
    if (const int prc = tome_plan(n, C, n_head, &p)) return prc == -2 ? STTM_ERR_UNSUPPORTED : STTM_ERR_ARG;
    if (workspace_bytes < p.total) return STTM_ERR_ARG;
    if (r < 1 || r > p.nb) return STTM_ERR_ARG;              // callers clamp r = min(r, n // 2) like the reference
    char* ws = reinterpret_cast<char*>(workspace);
    float* ahat = reinterpret_cast<float*>(ws + p.off_ahat);       // (16-bit inputs keep 16-bit unit rows in the same buffers)
    float* bhat = reinterpret_cast<float*>(ws + p.off_bhat);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ws + p.off_best);
    float* nmax = reinterpret_cast<float*>(ws + p.off_nmax);
    int* nidx = reinterpret_cast<int*>(ws + p.off_nidx);
    int* order = reinterpret_cast<int*>(ws + p.off_order);
    int* cnt = reinterpret_cast<int*>(ws + p.off_cnt);
    int* cur = reinterpret_cast<int*>(ws + p.off_cur);
    int* off = reinterpret_cast<int*>(ws + p.off_off);
    int* lists = reinterpret_cast<int*>(ws + p.off_lists);

    int* rank = reinterpret_cast<int*>(ws + p.off_parts);
    int* arrive = reinterpret_cast<int*>(ws + p.off_arrive);
    (void)hipMemsetAsync(best, 0, p.off_off - p.off_best, stream);     // best, cnt, cur, rank, arrive: one contiguous region
    const int ngrid = [&] { int g = (n + 3) / 4; return g > 8192 ? 8192 : g; }();
    // j-split: the grid is itiles*jsplit workgroups of ceil(jtiles/jsplit) tile products each, `resident` per CU.
    // Pick the split with the smallest per-CU critical path  ceil(WGs / slots) * tiles-per-WG  (a 588-WG grid on 512
    // slots leaves a third of the chip idle for the second round); ties go to the coarser split (fewer atomics).
    const int n_cu = tome_cu_count();
    auto pick_jsplit = [&](int itiles, int tj, int resident) {
        const int jtiles = (p.nb + tj - 1) / tj;
        int jsplit = 1;
        long best_cost = -1;
        for (int js = 1; js <= jtiles; ++js) {
            const long wgs = (long)itiles * js;
            const long per_cu = (wgs + n_cu - 1) / n_cu;
            long cost = per_cu * ((jtiles + js - 1) / js);
            if (per_cu < resident && js < jtiles) cost = cost * 3 / 2;       // a lone WG of a resident pair cannot hide its staging
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; jsplit = js; }
        }
        return jsplit;
    };
    // 256-tile kernels: all tile products spread evenly over one workgroup per CU (jsplit = 0, grid = it * js with js = 1 below)
    // when that shortens the per-CU critical path against the best per-a-tile split; "tome_flat" 0 = never, 2 = always (tests)
    auto pick_flat = [&](int& it, int& js) {
        const int mode = tome_flat_mode();
        const int jtiles = (p.nb + TG_T - 1) / TG_T;
        const long total = (long)it * jtiles;
        const long wgs = (long)it * js;
        const long split_cost = ((wgs + n_cu - 1) / n_cu) * ((jtiles + js - 1) / js);
        const long flat_wgs = total < n_cu ? total : n_cu;
        const long flat_cost = (total + flat_wgs - 1) / flat_wgs;
        // measured (tools/tome_ab_flat_prof.py): 69 x 69 tiles (T = 180) -- the best per-a-tile split is 4 761 one-product workgroups
        // in 19 rounds: bf16 686 -> 561 us, fp32 2 372 -> 2 277 us with 256 workgroups of 18 / 19 products; 49 x 49 (T = 128), where
        // the split fits one round (245 workgroups of 9 / 10): bf16 279 against 291 us flat -- so flat only replaces multi-round grids
        if (mode == 2 || (mode == 1 && (flat_cost < split_cost || (wgs > n_cu && flat_cost <= split_cost)))) { it = (int)flat_wgs; js = 0; }
    };
    const int itiles = (p.na + TM_I - 1) / TM_I;
    const int jsplit = pick_jsplit(itiles, TM_J, 2);
    const int split = tome_split_mode();
    if (dtype == STTM_F32 && split > 0) {
        uint16_t* ap = reinterpret_cast<uint16_t*>(ahat);
        uint16_t* bp = reinterpret_cast<uint16_t*>(bhat);
        if (n_head == 1 && C % 4 == 0 && reinterpret_cast<uintptr_t>(x_) % 16 == 0)
            hipLaunchKernelGGL(k_tome_normalize_split<4>, dim3(ngrid), dim3(256), 0, stream, reinterpret_cast<const float*>(x_), n, C,
                               n_head, p.D, p.Dp, ap, p.na, bp, p.nb);
        else
            hipLaunchKernelGGL(k_tome_normalize_split<1>, dim3(ngrid), dim3(256), 0, stream, reinterpret_cast<const float*>(x_), n, C,
                               n_head, p.D, p.Dp, ap, p.na, bp, p.nb);
        // tome_split: 1 = four product terms, 2 = three (without l.l); 3/4 force the 128-tile / the 256-tile DMA kernel (4 terms),
        // 5/6 the same with 3 terms.  The 256-tile kernel wins from ~6 k tokens on (measured cross-over: T = 32 frames of 196).
        // 7 = the four-wave form of the 256-tile kernel (128 x 128 wave tiles, round 6), 3 terms
        const int terms = (split == 2 || split == 5 || split == 6 || split == 7) ? 3 : 4;
        const bool big = split == 4 || split == 6 || split == 7 || ((split == 1 || split == 2) && p.na >= 3072);
        if (big) {
            int it = (p.na + TG_T - 1) / TG_T;
            int js = pick_jsplit(it, TG_T, 1);
            pick_flat(it, js);
#ifdef STTM_DEV
            const int abl = getenv("STTM_TOME_ABL") ? atoi(getenv("STTM_TOME_ABL")) : 0;
            if (abl == 1) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t, 1>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 2) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t, 2>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 3) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t, 3>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 4) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t, 4>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 5) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t, 5>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 6) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t, 6>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 8 && split == 7) hipLaunchKernelGGL((k_tome_match_glds<2, 3, f16_t, 8, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (abl == 8) hipLaunchKernelGGL((k_tome_match_glds<2, 3, f16_t, 8>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else
#endif
            if (split == 7) hipLaunchKernelGGL((k_tome_match_glds<2, 3, f16_t, 0, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else if (terms == 4) hipLaunchKernelGGL((k_tome_match_glds<2, 4, f16_t>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
            else hipLaunchKernelGGL((k_tome_match_glds<2, 3, f16_t>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best);
        } else {
            const size_t lds = (size_t)(2 * TM_I + 2 * TM_J) * (TM_K + 8) * 2;
            if (terms == 4) hipLaunchKernelGGL((k_tome_match_split<TM_K, 2, 2, 4>), dim3(itiles * jsplit), dim3(256), lds, stream, ap, bp, p.na, p.nb, p.Dp, jsplit, best);
            else hipLaunchKernelGGL((k_tome_match_split<TM_K, 2, 2, 3>), dim3(itiles * jsplit), dim3(256), lds, stream, ap, bp, p.na, p.nb, p.Dp, jsplit, best);
        }
    } else if (dtype == STTM_F32) {
        hipLaunchKernelGGL(k_tome_normalize, dim3(ngrid), dim3(256), 0, stream, reinterpret_cast<const float*>(x_), n, C, n_head, p.D, p.Dp, ahat, bhat);
        hipLaunchKernelGGL(k_tome_match, dim3(itiles * jsplit), dim3(256), 0, stream, ahat, bhat, p.na, p.nb, p.Dp, jsplit, best);
    } else {
        // 16-bit inputs: the 256-tile DMA kernel from ~6 k tokens on ("tome_split" 3 / 5 force the 128-tile kernel, 4 / 6 the
        // 256-tile one), like the split path
        uint16_t* ap = reinterpret_cast<uint16_t*>(ahat);
        uint16_t* bp = reinterpret_cast<uint16_t*>(bhat);
        const bool big = split == 4 || split == 6 || split == 7 || (split != 3 && split != 5 && p.na >= 3072);
        int it = (p.na + TG_T - 1) / TG_T;
        int js = pick_jsplit(it, TG_T, 1);
        pick_flat(it, js);
#ifdef STTM_DEV
        const int abl16 = getenv("STTM_TOME_ABL") ? atoi(getenv("STTM_TOME_ABL")) : 0;
        constexpr int ABL16A = 3, ABL16B = 4, ABL16C = 5, ABL4A = 1, ABL4B = 5, ABL4C = 6, ABL4D = 7, ABLSW = 8;      // (four-wave form: 1 no DMA, 5 MFMAs alone, 6 + fragment reads, 7 no running max; 8: the one-pass sweep, either form)
#else
        constexpr int abl16 = 0, ABL16A = 0, ABL16B = 0, ABL16C = 0, ABL4A = 0, ABL4B = 0, ABL4C = 0, ABL4D = 0, ABLSW = 0;
#endif
#define STTM_TOME_16(TT)                                                                                                            \
        do {                                                                                                                        \
            if (n_head == 1 && C % 8 == 0 && C <= 4096 && reinterpret_cast<uintptr_t>(x_) % 16 == 0)                              \
                hipLaunchKernelGGL((k_tome_normalize16<TT, 8>), dim3(ngrid), dim3(256), 0, stream, x_, n, C, n_head, p.D, p.Dp, ap, bp); \
            else                                                                                                                    \
                hipLaunchKernelGGL((k_tome_normalize16<TT, 1>), dim3(ngrid), dim3(256), 0, stream, x_, n, C, n_head, p.D, p.Dp, ap, bp); \
            if (big && abl16 == 3) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL16A>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && abl16 == 4) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL16B>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && abl16 == 5) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL16C>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && split == 7 && abl16 == 8) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABLSW, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && abl16 == 8) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABLSW>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && split == 7 && abl16 == 1) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL4A, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && split == 7 && abl16 == 5) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL4B, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && split == 7 && abl16 == 6) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL4C, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && split == 7 && abl16 == 7) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, ABL4D, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big && split == 7) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT, 0, 4>), dim3(js ? it * js : it), dim3(256), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else if (big) hipLaunchKernelGGL((k_tome_match_glds<1, 1, TT>), dim3(js ? it * js : it), dim3(512), 2 * TG_BUF, stream, ap, bp, p.na, p.nb, p.Dp, js, best); \
            else hipLaunchKernelGGL(k_tome_match16<TT>, dim3(itiles * jsplit), dim3(256), 0, stream, ap, bp, p.na, p.nb, p.Dp, jsplit, best); \
        } while (0)
        if (dtype == STTM_BF16) STTM_TOME_16(bf16_t); else STTM_TOME_16(f16_t);
#undef STTM_TOME_16
    }
    {
        // ranking by counting (+ the scan, in its last workgroup) and the list fill (round 4): 2 launches
        const int iblocks = (p.na + RK_I - 1) / RK_I;
        int js = (8 * n_cu + iblocks - 1) / iblocks;                    // ~8 four-wave workgroups per CU: the scalar key loads of a wave
        const int max_js = (p.na + 255) / 256;                          // are not pipelined, other waves cover them
        if (js > max_js) js = max_js;
        if (js > kRankParts) js = kRankParts;
        if (js < 1) js = 1;
        hipLaunchKernelGGL(k_tome_rank, dim3(iblocks * js), dim3(RK_I), 0, stream, best, p.na, js, r, rank, arrive, nmax, nidx, order, cnt, p.nb, off);
        hipLaunchKernelGGL(k_tome_fill, dim3((r + 255) / 256), dim3(256), 0, stream, order, nidx, r, off, cur, lists);
    }
    {
        int grid = (n - r + 3) / 4; if (grid > 8192) grid = 8192;
#define STTM_TOME_MERGE(TT, VV) hipLaunchKernelGGL((k_tome_merge<TT, VV>), dim3(grid), dim3(256), 0, stream, x_, size, idx, n, C, p.na, p.nb, r, order, off, lists, x_out_, size_out, idx_out)
        const size_t eb = dtype == STTM_F32 ? 4 : 2;
        const bool v4 = C % 4 == 0 && reinterpret_cast<uintptr_t>(x_) % (4 * eb) == 0 && reinterpret_cast<uintptr_t>(x_out_) % (4 * eb) == 0;
        const bool v8 = C % 8 == 0 && reinterpret_cast<uintptr_t>(x_) % 16 == 0 && reinterpret_cast<uintptr_t>(x_out_) % 16 == 0;
        if (dtype == STTM_F32) { if (v4) STTM_TOME_MERGE(float, 4); else STTM_TOME_MERGE(float, 1); }
        else if (dtype == STTM_BF16) { if (v8) STTM_TOME_MERGE(bf16_t, 8); else if (v4) STTM_TOME_MERGE(bf16_t, 4); else STTM_TOME_MERGE(bf16_t, 1); }
        else { if (v8) STTM_TOME_MERGE(f16_t, 8); else if (v4) STTM_TOME_MERGE(f16_t, 4); else STTM_TOME_MERGE(f16_t, 1); }
#undef STTM_TOME_MERGE
    }
    if (node_max_out) (void)hipMemcpyAsync(node_max_out, nmax, (size_t)p.na * 4, hipMemcpyDeviceToDevice, stream);
    if (node_idx_out) (void)hipMemcpyAsync(node_idx_out, nidx, (size_t)p.na * 4, hipMemcpyDeviceToDevice, stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STTM_OK : STTM_ERR_LAUNCH;
}

}  // extern "C"
