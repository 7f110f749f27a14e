// Probe: does hipExtAnyOrderLaunch let kernel B of the SAME stream start while kernel A still runs (gfx950, ROCm 7.2)?
// and are all workgroups of A placed before the first workgroup of B?   Build: hipcc --offload-arch=gfx950 -O2 -o anyorder_probe anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }
__global__ void kA(long long* start, long long* end, int* flag, long long spin_ticks, int* xcc) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; xcc[blockIdx.x] = xcc_id(); }
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0) {
        end[blockIdx.x] = wall_clock64();
        __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void kB(long long* start, long long* seen, int* flag, int target, int* xcc) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        start[blockIdx.x] = t0; xcc[blockIdx.x] = xcc_id();
        int ok = 0;
        while (wall_clock64() - t0 < 100000000ll) {   // 1 s
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        seen[blockIdx.x] = ok ? wall_clock64() : -1;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int maxA = 8192, maxB = 8192;
    long long *sa, *ea, *sb, *nb; int* flag; int *xa, *xb; CK(hipMalloc(&xa, maxA * 4)); CK(hipMalloc(&xb, maxB * 4));
    CK(hipMalloc(&sa, maxA * 8)); CK(hipMalloc(&ea, maxA * 8)); CK(hipMalloc(&sb, maxB * 8)); CK(hipMalloc(&nb, maxB * 8)); CK(hipMalloc(&flag, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Cfg { int ga, ta, gb, tb; long long spin; int any; };
    Cfg cfgs[] = {
        {256, 256, 64, 256, 2000, 0}, {256, 256, 64, 256, 2000, 1},          // A fits: 20 us spin
        {4096, 1024, 64, 256, 500, 0}, {4096, 1024, 64, 256, 500, 1},        // A = 8 rounds of 5 us
        {16, 1024, 4096, 256, 1500, 1},                                      // label-like A, machine-filling B
        {2048, 256, 16, 1024, 1000, 1}, {2048, 256, 16, 1024, 1000, 0}, {2048, 256, 128, 256, 1000, 1}, {16, 1024, 2048, 256, 1500, 0},
    };
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(flag, 0, 4, st));
            CK(hipMemsetAsync(sb, 0, maxB * 8, st));
            CK(hipStreamSynchronize(st));
            hipLaunchKernelGGL(kA, dim3(c.ga), dim3(c.ta), 0, st, sa, ea, flag, c.spin, xa);
            if (c.any) hipExtLaunchKernelGGL(kB, dim3(c.gb), dim3(c.tb), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, sb, nb, flag, c.ga, xb);
            else hipLaunchKernelGGL(kB, dim3(c.gb), dim3(c.tb), 0, st, sb, nb, flag, c.ga, xb);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            std::vector<long long> hsa(c.ga), hea(c.ga), hsb(c.gb), hnb(c.gb);
            CK(hipMemcpy(hsa.data(), sa, c.ga * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hea.data(), ea, c.ga * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hsb.data(), sb, c.gb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hnb.data(), nb, c.gb * 8, hipMemcpyDeviceToHost));
            std::vector<int> hxa(c.ga), hxb(c.gb);
            CK(hipMemcpy(hxa.data(), xa, c.ga * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hxb.data(), xb, c.gb * 4, hipMemcpyDeviceToHost));
            const long long a0 = *std::min_element(hsa.begin(), hsa.end()), a_last_start = *std::max_element(hsa.begin(), hsa.end());
            const long long a_end = *std::max_element(hea.begin(), hea.end());
            const long long b0 = *std::min_element(hsb.begin(), hsb.end()), b_last = *std::max_element(hsb.begin(), hsb.end());
            const long long seen_min = *std::min_element(hnb.begin(), hnb.end()), seen_max = *std::max_element(hnb.begin(), hnb.end());
            printf("A %dx%d spin %.1fus  B %dx%d any=%d | (us from A's first start) A last start %.2f  A end %.2f | B first start %.2f last start %.2f | flag seen %.2f .. %.2f%s\n",
                   c.ga, c.ta, c.spin / 100.0, c.gb, c.tb, c.any, (a_last_start - a0) / 100.0, (a_end - a0) / 100.0, (b0 - a0) / 100.0,
                   (b_last - a0) / 100.0, (seen_min - a0) / 100.0, (seen_max - a0) / 100.0, seen_min < 0 ? "  TIMEOUT" : "");
            if (rep == 2 && c.any) for (int x = 0; x < 8; ++x) {
                long long ae = 0, als = 0, bs0 = 1ll << 62, bs1 = 0; int nb_ = 0;
                for (int i = 0; i < c.ga; ++i) if (hxa[i] == x) { ae = std::max(ae, hea[i]); als = std::max(als, hsa[i]); }
                for (int i = 0; i < c.gb; ++i) if (hxb[i] == x) { bs0 = std::min(bs0, hsb[i]); bs1 = std::max(bs1, hsb[i]); ++nb_; }
                printf("    xcd %d: A last start %.2f end %.2f | B (%d wgs) start %.2f .. %.2f\n", x, (als - a0) / 100.0, (ae - a0) / 100.0, nb_, (bs0 - a0) / 100.0, (bs1 - a0) / 100.0);
            }
        }
    }
    // chain of 4: normal vs any-order for kernels 2..4, wall time per chain
    for (int any = 0; any < 2; ++any) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int n = 200;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(kA, dim3(64), dim3(256), 0, st, sa, ea, flag, 10ll, xa);
            for (int k = 0; k < 3; ++k) {
                if (any) hipExtLaunchKernelGGL(kA, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, sa, ea, flag, 10ll, xa);
                else hipLaunchKernelGGL(kA, dim3(64), dim3(256), 0, st, sa, ea, flag, 10ll, xa);
            }
        }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("chain of 4 tiny kernels, any=%d: %.2f us per chain\n", any, ms * 1000 / n);
    }
    return 0;
}
