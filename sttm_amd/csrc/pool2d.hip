// Upstream 2-D pooling of the projected vision tokens: replaces
// LlavaMetaForCausalLM.get_2dPool (llava/model/llava_arch.py:173-198 of the reference):
//     [T, side*side, C] -> view [T, side, side, C] -> permute/contiguous NCHW -> avg_pool2d / max_pool2d / bilinear
//     F.interpolate(size=ceil(side/stride)) -> permute back -> [T, out*out, C]
// Here the tokens stay channels-last the whole time: one thread owns VEC consecutive channels of one OUTPUT token,
// reads the 2x2 (bilinear) or stride x stride (average / max) source tokens with 16-byte loads and writes one pack.
// HBM-bound: read T*side^2*C once (neighbouring outputs share source rows through L2), write T*out^2*C.
// Arithmetic is float32 with one rounding per operation and no contraction, in the order of oracle/pool_oracle.py;
// the bilinear source index uses ONE fused multiply-add like the ATen builds (see the oracle's note on index 8 of 27->14).
#include "sttm_common.h"
#include "sttm_kernels.h"

// one rounding per float operation in this file (plain operators, NOT the __fmul_rn/__fadd_rn wrappers: those are
// compiled with contraction allowed in the HIP headers and fuse again once inlined)
#pragma clang fp contract(off)

namespace sttm {

struct PoolArgs {
    const void* x; void* out;
    int T, H, W, C, OH, OW, stride, mode;
    float sh, sw;
};

// Round 6: ONE WORKGROUP PER OUTPUT TOKEN, a thread per 16-byte pack of its channels.  The token's frame / row / column and the source
// taps are workgroup-uniform (scalar registers, computed once); a thread issues its four (bilinear) source loads, blends and stores.
// Rounds 2-5 ran a flat grid-stride loop over (token, pack) pairs: four 64-bit integer divisions per 16-byte store (idx % P, idx / P,
// tok % OW, ...: ~1000 instructions around 4 loads), bf16 C = 3584 at 4.5 TB/s.  Same per-element arithmetic, same bits.
template <typename T, int VEC>
__global__ void __launch_bounds__(512) k_pool2d(PoolArgs a) {
    const int P = a.C / VEC;
    const int tok = blockIdx.x;
    const int ox = tok % a.OW, rest = tok / a.OW;
    const int oy = rest % a.OH, t = rest / a.OH;
    const int64_t frame = (int64_t)t * a.H * a.W;
    if (a.mode == STTM_POOL_NEAREST) {
        // F.interpolate(size=...) default mode: src = min(floor(dst * (in / out)), in - 1), float32 product
        int sy = (int)floorf((float)oy * a.sh), sx = (int)floorf((float)ox * a.sw);
        sy = sy < a.H - 1 ? sy : a.H - 1;
        sx = sx < a.W - 1 ? sx : a.W - 1;
        const int64_t src = (frame + (int64_t)sy * a.W + sx) * a.C;
        for (int pk = threadIdx.x; pk < P; pk += blockDim.x)
            store_pack<T, VEC>(a.out, (int64_t)tok * a.C + pk * VEC, load_pack<T, VEC>(a.x, src + pk * VEC));
    } else if (a.mode == STTM_POOL_BILINEAR) {
        int y0, y1, x0, x1;
        float h0, h1, w0, w1;
        bilinear_tap(a.sh, oy, a.H, y0, y1, h0, h1);
        bilinear_tap(a.sw, ox, a.W, x0, x1, w0, w1);
        const int64_t s00 = (frame + (int64_t)y0 * a.W + x0) * a.C, s01 = (frame + (int64_t)y0 * a.W + x1) * a.C;
        const int64_t s10 = (frame + (int64_t)y1 * a.W + x0) * a.C, s11 = (frame + (int64_t)y1 * a.W + x1) * a.C;
        for (int pk = threadIdx.x; pk < P; pk += blockDim.x) {
            const int c0 = pk * VEC;
            const Pack<T, VEC> p00 = load_pack<T, VEC>(a.x, s00 + c0), p01 = load_pack<T, VEC>(a.x, s01 + c0);
            const Pack<T, VEC> p10 = load_pack<T, VEC>(a.x, s10 + c0), p11 = load_pack<T, VEC>(a.x, s11 + c0);
            Pack<T, VEC> res;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float top = w0 * p00.get(e) + w1 * p01.get(e);        // contraction is off in this file
                const float bot = w0 * p10.get(e) + w1 * p11.get(e);
                res.set(e, h0 * top + h1 * bot);
            }
            store_pack<T, VEC>(a.out, (int64_t)tok * a.C + c0, res);
        }
    } else {
        const bool is_max = a.mode == STTM_POOL_MAX;
        const float den = (float)(a.stride * a.stride);
        for (int pk = threadIdx.x; pk < P; pk += blockDim.x) {
            const int c0 = pk * VEC;
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = is_max ? -INFINITY : 0.f;
            for (int dy = 0; dy < a.stride; ++dy)
                for (int dx = 0; dx < a.stride; ++dx) {
                    const Pack<T, VEC> p = load_pack<T, VEC>(a.x, (frame + (int64_t)(oy * a.stride + dy) * a.W + (ox * a.stride + dx)) * a.C + c0);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float v = p.get(e);
                        acc[e] = is_max ? (v > acc[e] ? v : acc[e]) : acc[e] + v;
                    }
                }
            Pack<T, VEC> res;
#pragma unroll
            for (int e = 0; e < VEC; ++e) res.set(e, is_max ? acc[e] : acc[e] / den);
            store_pack<T, VEC>(a.out, (int64_t)tok * a.C + c0, res);
        }
    }
}

template <typename T>
static hipError_t launch_pool_t(const PoolArgs& a, int vec, hipStream_t stream) {
    const int64_t tokens = (int64_t)a.T * a.OH * a.OW;
    if (tokens < 1 || tokens > 0x7fffffff) return hipErrorInvalidValue;
    int nt = (a.C / vec + 63) / 64 * 64;                       // a thread per pack, whole waves, at most 512 (wider rows loop)
    if (nt > 512) nt = 512;
    const dim3 grid((unsigned)tokens), block((unsigned)nt);
    if constexpr (TypeInfo<T>::lowp) {
        if (vec == 8) hipLaunchKernelGGL((k_pool2d<T, 8>), grid, block, 0, stream, a);
        else if (vec == 4) hipLaunchKernelGGL((k_pool2d<T, 4>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_pool2d<T, 2>), grid, block, 0, stream, a);
    } else {
        if (vec == 4) hipLaunchKernelGGL((k_pool2d<T, 4>), grid, block, 0, stream, a);
        else if (vec == 2) hipLaunchKernelGGL((k_pool2d<T, 2>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_pool2d<T, 1>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_pool2d(const void* x, void* out, int T, int H, int W, int C, int OH, int OW, int stride, int mode, int dtype,
                         hipStream_t stream) {
    PoolArgs a;
    a.x = x; a.out = out; a.T = T; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.stride = stride; a.mode = mode;
    a.sh = (float)H / (float)OH; a.sw = (float)W / (float)OW;
    const int eb = dtype == STTM_F32 ? 4 : 2;
    const uintptr_t align = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
    int vec = 16 / eb;                                         // 16-byte packs when the rows allow it
    while (vec > (eb == 4 ? 1 : 2) && (C % vec || (align % (vec * eb)) || ((int64_t)C * eb) % (vec * eb))) vec >>= 1;
    if (eb == 2 && (C % 2)) return hipErrorInvalidValue;      // 16-bit rows are handled in pairs
    if (dtype == STTM_F32) return launch_pool_t<float>(a, vec, stream);
    if (dtype == STTM_BF16) return launch_pool_t<bf16_t>(a, vec, stream);
    return launch_pool_t<f16_t>(a, vec, stream);
}

}  // namespace sttm
