// What does COLD straight-line code cost on gfx950?  (Round 6: the fused label kernel k_col_labels<2> is 71 KB of code for 14 us of work on 16
// workgroups; the ToMe ranking kernel is a chain of fixed latencies.)  Kernels of N KB of straight-line VALU code (v_add_f32 with an inline
// constant: 4 bytes each) executed ONCE per wave, against the same instruction count run as a loop over a 1 KB body; 16 workgroups x 1024 threads
// (the label stage's shape) or 1 wave; back to back (code warm in the L2) or behind a kernel that streams 1 GB (code pushed out of the L2s).
// In-kernel stamps (wall_clock64, 100 MHz) around the body: max(end) - min(start) over the workgroups; HIP-event time of the launch beside it.
//   hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip && ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define ADD4 "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
#define ADD16 ADD4 ADD4 ADD4 ADD4
#define ADD64 ADD16 ADD16 ADD16 ADD16
#define ADD256 ADD64 ADD64 ADD64 ADD64            /* 256 instructions = 1 KB */

template <int KB> __device__ __forceinline__ float straight(float x) {
    if constexpr (KB > 0) {
        asm volatile(ADD256 : "+v"(x));
        return straight<KB - 1>(x);
    } else {
        return x;
    }
}

template <int KB, bool LOOP>
__global__ void __launch_bounds__(1024) k_code(float* out, unsigned long long* stamps) {
    float x = (float)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    if constexpr (LOOP) {
#pragma unroll 1
        for (int i = 0; i < KB; ++i) asm volatile(ADD256 : "+v"(x));
    } else {
        x = straight<KB>(x);
    }
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

__global__ void k_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i]; v.x += 1.f; out[i] = v;
    }
}

template <int KB, bool LOOP>
static void run(const char* what, int wgs, int threads, bool flush, float* out, unsigned long long* stamps, float4* big_in, float4* big_out, size_t big_n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> body, ev;
    std::vector<unsigned long long> h(2 * wgs);
    for (int rep = 0; rep < 12; ++rep) {
        if (flush) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, big_in, big_out, big_n);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_code<KB, LOOP>), dim3(wgs), dim3(threads), 0, 0, out, stamps);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < wgs; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
        if (rep >= 2) { body.push_back((hi - lo) * 0.01); ev.push_back(ms * 1e3); }
    }
    std::sort(body.begin(), body.end()); std::sort(ev.begin(), ev.end());
    printf("| %3d KB %-8s | %2d x %4d | %-22s | %7.2f | %7.2f |\n", KB, LOOP ? "loop" : "straight", wgs, threads, what, body[body.size() / 2], ev[ev.size() / 2]);
}

int main() {
    float* out; unsigned long long* stamps; float4 *bi, *bo;
    const size_t big_n = (512ull << 20) / sizeof(float4);
    hipMalloc(&out, 64 * 1024 * sizeof(float)); hipMalloc(&stamps, 256 * sizeof(unsigned long long));
    hipMalloc(&bi, big_n * sizeof(float4)); hipMalloc(&bo, big_n * sizeof(float4));
    hipMemset(bi, 0, big_n * sizeof(float4));
    printf("| code | grid | launched | body us (stamps: first start to last end) | launch us (events) |\n|---|---|---|---|---|\n");
#define BOTH(KB, WGS, THR)                                                                          \
    run<KB, false>("back to back", WGS, THR, false, out, stamps, bi, bo, big_n);                     \
    run<KB, true>("back to back", WGS, THR, false, out, stamps, bi, bo, big_n);                      \
    run<KB, false>("behind a 1 GB stream", WGS, THR, true, out, stamps, bi, bo, big_n);             \
    run<KB, true>("behind a 1 GB stream", WGS, THR, true, out, stamps, bi, bo, big_n);
    BOTH(4, 16, 1024) BOTH(16, 16, 1024) BOTH(32, 16, 1024) BOTH(64, 16, 1024) BOTH(96, 16, 1024)
    BOTH(32, 1, 64) BOTH(32, 2048, 256)
    return 0;
}
