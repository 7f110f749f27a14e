// Shared device/host helpers for the STTM gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libsttm_hip is written for gfx950 (MI355X) only: wave64, the gfx950 MFMA / permlane-swap / LDS-DMA instructions and the sc1 hand-off forms its kernels rely on"
#endif

namespace sttm {

constexpr int kMaxLevels = 6;   // deepest pyramid the fused spatial kernel supports (root .. leaf): root cells of up to 32 x 32 leaves
constexpr int kWave = 64;

// ---- geometry (closed form of quadtree_spatial_merger.py:155-271 of the reference) -----------------
// Children of parent index i along one axis when a side of n_child cells is halved:
//   even: [2i, 2i+1]        odd: i == 0 -> [0]   else [2i-1, 2i]      (first cell stays alone)
__host__ __device__ __forceinline__ int child_start(int i, int n_child) {
    return (n_child & 1) ? (i == 0 ? 0 : 2 * i - 1) : 2 * i;
}
__host__ __device__ __forceinline__ int child_count(int i, int n_child) {
    return ((n_child & 1) && i == 0) ? 1 : 2;
}

// inverse of child_start / child_count: the parent index of child c along an axis whose child level has n_child cells
__host__ __device__ __forceinline__ int parent_of(int c, int n_child) {
    return (n_child & 1) ? (c == 0 ? 0 : (c + 1) / 2) : c / 2;
}

struct LevelDims {
    int n_level;              // pyramid levels built, root level first (index 0), leaf last
    int h[kMaxLevels + 1];
    int w[kMaxLevels + 1];
};

__host__ __device__ __forceinline__ constexpr int pow4(int d) { return 1 << (2 * d); }
// id of the first node at tree depth d in a complete 4-ary tree stored level by level
__host__ __device__ __forceinline__ constexpr int depth_base(int d) { return (pow4(d) - 1) / 3; }

// ---- storage packs: VEC consecutive channels of one token, kept in the INPUT dtype -----------------
struct bf16_t { uint16_t x; };
struct f16_t { uint16_t x; };

__device__ __forceinline__ float bf16_bits_to_float(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ uint32_t float_to_bf16_bits(float f) {   // round to nearest even, NaN stays NaN
    // gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32); the integer emulation cost ~6 VALU instructions per element,
    // which made the 16-bit pooling / group-sum paths VALU-bound (1000+ instructions per thread in the spatial kernel)
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float f16_bits_to_float(uint32_t b) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)b);
}
__device__ __forceinline__ uint32_t float_to_f16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);
}

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> { static constexpr int bytes = 4; static constexpr bool lowp = false; };
template <> struct TypeInfo<bf16_t> { static constexpr int bytes = 2; static constexpr bool lowp = true; };
template <> struct TypeInfo<f16_t> { static constexpr int bytes = 2; static constexpr bool lowp = true; };

template <typename T, int VEC> struct Pack;

template <int VEC> struct alignas(4 * VEC) Pack<float, VEC> {
    float v[VEC];
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = f; }
    __device__ __forceinline__ void set_pair(int i, float f0, float f1) { v[i] = f0; v[i + 1] = f1; }      // i even
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = 0.f;
    }
};

template <typename T16, int VEC> struct alignas(2 * VEC) Pack16 {
    static_assert(VEC % 2 == 0, "16-bit packs hold an even number of channels");
    uint32_t w[VEC / 2];
    __device__ __forceinline__ uint32_t bits(int i) const { return (i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu); }
    __device__ __forceinline__ void set_bits(int i, uint32_t b) {
        if (i & 1) w[i >> 1] = (w[i >> 1] & 0x0000ffffu) | (b << 16);
        else w[i >> 1] = (w[i >> 1] & 0xffff0000u) | (b & 0xffffu);
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < VEC / 2; ++i) w[i] = 0u;
    }
};
typedef float sttm_cvt_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sttm_cvt_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sttm_cvt_f16x2 __attribute__((ext_vector_type(2)));
template <int VEC> struct Pack<bf16_t, VEC> : Pack16<bf16_t, VEC> {
    __device__ __forceinline__ float get(int i) const { return bf16_bits_to_float(this->bits(i)); }
    __device__ __forceinline__ void set(int i, float f) { this->set_bits(i, float_to_bf16_bits(f)); }
    // two adjacent channels (i even) with ONE v_cvt_pk_bf16_f32 and no merge of 16-bit halves
    __device__ __forceinline__ void set_pair(int i, float f0, float f1) {
        const sttm_cvt_f32x2 f = {f0, f1};
        this->w[i >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, sttm_cvt_bf16x2));
    }
};
template <int VEC> struct Pack<f16_t, VEC> : Pack16<f16_t, VEC> {
    __device__ __forceinline__ float get(int i) const { return f16_bits_to_float(this->bits(i)); }
    __device__ __forceinline__ void set(int i, float f) { this->set_bits(i, float_to_f16_bits(f)); }
    __device__ __forceinline__ void set_pair(int i, float f0, float f1) {
        const sttm_cvt_f32x2 f = {f0, f1};
        this->w[i >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, sttm_cvt_f16x2));
    }
};
// out[e] = fn(e) for every channel of a pack, written pairwise where the pack allows it
template <typename T, int VEC, typename F>
__device__ __forceinline__ void pack_fill(Pack<T, VEC>& out, F fn) {
    if constexpr (VEC % 2 == 0) {
#pragma unroll
        for (int e = 0; e < VEC; e += 2) out.set_pair(e, fn(e), fn(e + 1));
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) out.set(e, fn(e));
    }
}

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack(const void* base, int64_t elem_off) {
    return *reinterpret_cast<const Pack<T, VEC>*>(reinterpret_cast<const char*>(base) + elem_off * TypeInfo<T>::bytes);
}
// streaming variants (read-once / write-once data): nontemporal hint, keeps the L2 for the re-read structures
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack_stream(const void* base, int64_t elem_off) {
#ifdef STTM_NT_STREAM
    const char* p = reinterpret_cast<const char*>(base) + elem_off * TypeInfo<T>::bytes;
    Pack<T, VEC> out;
    constexpr int bytes = TypeInfo<T>::bytes * VEC;
    if constexpr (bytes == 16) {
        const auto v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((ext_vector_type(4))) unsigned*>(p));
        __builtin_memcpy(&out, &v, 16);
    } else if constexpr (bytes == 8) {
        const auto v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned*>(p));
        __builtin_memcpy(&out, &v, 8);
    } else if constexpr (bytes == 4) {
        const unsigned v = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p));
        __builtin_memcpy(&out, &v, 4);
    } else {
        out = *reinterpret_cast<const Pack<T, VEC>*>(p);
    }
    return out;
#else
    return load_pack<T, VEC>(base, elem_off);
#endif
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack_stream(void* base, int64_t elem_off, const Pack<T, VEC>& p) {
#ifdef STTM_NT_STREAM
    char* d = reinterpret_cast<char*>(base) + elem_off * TypeInfo<T>::bytes;
    constexpr int bytes = TypeInfo<T>::bytes * VEC;
    if constexpr (bytes == 16) {
        __attribute__((ext_vector_type(4))) unsigned v;
        __builtin_memcpy(&v, &p, 16);
        __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(d));
    } else if constexpr (bytes == 8) {
        __attribute__((ext_vector_type(2))) unsigned v;
        __builtin_memcpy(&v, &p, 8);
        __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((ext_vector_type(2))) unsigned*>(d));
    } else {
        *reinterpret_cast<Pack<T, VEC>*>(d) = p;
    }
#else
    *reinterpret_cast<Pack<T, VEC>*>(reinterpret_cast<char*>(base) + elem_off * TypeInfo<T>::bytes) = p;
#endif
}

template <typename T, int VEC>
__device__ __forceinline__ void store_pack(void* base, int64_t elem_off, const Pack<T, VEC>& p) {
    *reinterpret_cast<Pack<T, VEC>*>(reinterpret_cast<char*>(base) + elem_off * TypeInfo<T>::bytes) = p;
}

template <typename T, int VEC>
__device__ __forceinline__ float dot_pack(const Pack<T, VEC>& a, const Pack<T, VEC>& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s = fmaf(a.get(i), b.get(i), s);
    return s;
}
// fp32, even widths: two independent fma chains on a float2 (v_pk_fma_f32: two fmas per VALU issue), joined at the end
typedef float sttm_f32x2 __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ typename std::enable_if<(VEC % 2 == 0), float>::type
dot_pack(const Pack<float, VEC>& a, const Pack<float, VEC>& b) {
    sttm_f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VEC; i += 2) {
        const sttm_f32x2 x = {a.v[i], a.v[i + 1]}, y = {b.v[i], b.v[i + 1]};
        acc = __builtin_elementwise_fma(x, y, acc);
    }
    return acc.x + acc.y;
}

// 16-bit inputs: packed dot-product instructions (v_dot2c_f32_bf16 / v_dot2c_f32_f16): two exact products added into an
// fp32 accumulator per instruction, no unpacking.  Half the multiply-adds of the fp32 path and none of its conversions.
typedef __bf16 sttm_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sttm_f16x2 __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ float dot_pack(const Pack<bf16_t, VEC>& a, const Pack<bf16_t, VEC>& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i)
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sttm_bf16x2, a.w[i]), __builtin_bit_cast(sttm_bf16x2, b.w[i]), s, false);
    return s;
}
template <int VEC>
__device__ __forceinline__ float dot_pack(const Pack<f16_t, VEC>& a, const Pack<f16_t, VEC>& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i)
        s = __builtin_amdgcn_fdot2(__builtin_bit_cast(sttm_f16x2, a.w[i]), __builtin_bit_cast(sttm_f16x2, b.w[i]), s, false);
    return s;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)); this one lets global traffic stay in flight across the barrier.  Use it only
// where the data exchanged through the barrier lives in LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- wave-level reductions -----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Butterfly "transpose-reduce": every lane enters with N partial values, and leaves with the full
// 64-lane total of ONE of them (value index reduce_slot<N>(lane); lanes whose slot is padding get
// reduce_valid == false).  Costs ~N shuffles instead of 6*N.
template <int N, int SWAP = 0>      // SWAP: 0 = selects + ds_bpermute at every stage, 1 = lane swap at stage 32, 2 = at stages 32 and 16,
                                    //       3 = the same through in-place inline asm (no copies for the tied operands: the 128-register fp32 kernel)
__device__ __forceinline__ float wave_reduce_many(float (&v)[N], int lane) {
    static_assert(N >= 1 && N <= 64, "one result per lane");
    int n = N;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int half = (n + 1) / 2;
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float lo = v[i];
            const float hi = (i + half < n) ? v[i + half] : 0.f;
            if constexpr (SWAP > 0) {
            // SWAP (16-bit kernels, where the registers allow it -- the 128-register fp32 kernel spills 52 bytes with it):
            // stages 32 and 16 as ONE lane-swap instruction + one add per output instead of two selects + a ds_bpermute + an add:
            // v_permlane32_swap exchanges the upper half of `lo` with the lower half of `hi`, v_permlane16_swap the odd 16-lane
            // rows of `lo` with the even rows of `hi`; afterwards the two registers hold exactly (own kept value, partner's sent
            // value) in every lane -- the same two operands as the select form, so the sums are bit-identical.
            if (m == 32) {
                if constexpr (SWAP >= 3) {            // in place (inline asm: no copies for the tied operands)
                    float a = lo, b = hi;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));     // (the hazard recogniser does not see inside: wait states by hand)
                    v[i] = a + b;
                } else {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
                continue;
            }
            if (m == 16 && SWAP > 1) {
                if constexpr (SWAP >= 3) {
                    float a = lo, b = hi;
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
                    v[i] = a + b;
                } else {
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
                continue;
            }
            }
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            v[i] = keep + __shfl_xor(send, m, 64);
        }
        n = half;
    }
    return v[0];
}
template <int N>
__device__ __forceinline__ int wave_reduce_slot(int lane, bool& valid) {
    int n = N, nv = N, idx = 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int half = (n + 1) / 2;
        if (lane & m) { idx += half; nv = nv - half; if (nv < 0) nv = 0; }
        else { if (nv > half) nv = half; }
        n = half;
    }
    valid = nv >= 1;
    return idx;
}

// Source taps of one output coordinate of F.interpolate(mode="bilinear", align_corners=False) (ATen's area_pixel_compute_source_index +
// guard): shared by the stand-alone pooling kernel (pool2d.hip) and the spatial kernel's fused pooled-leaf load.  One rounding per
// operation, the source index through ONE fused multiply-add like the ATen builds (oracle/pool_oracle.py's note on index 8 of 27 -> 14).
__device__ __forceinline__ void bilinear_tap(float scale, int o, int n_in, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)
    float src = __builtin_fmaf(scale, (float)o + 0.5f, -0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

}  // namespace sttm
