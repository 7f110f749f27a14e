// Read-only / write-only / copy bandwidth of MI355X at the sizes of the STTM pipeline (rows of 4 KB, 16 B per lane).
// Build: hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_read(const f4* __restrict__ x, size_t n, f4* sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc.x == 12345.678f) sink[0] = acc;
}
__global__ void k_write(f4* __restrict__ y, size_t n) {
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = v;
}
__global__ void k_copy(const f4* __restrict__ x, f4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const size_t big = 1ull << 30;
    f4 *a, *b, *flush;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&flush, big));
    CK(hipMemset(a, 1, big)); CK(hipMemset(b, 0, big)); CK(hipMemset(flush, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {46ull << 20, 103ull << 20, 512ull << 20};
    for (size_t bytes : sizes) {
        const size_t n = bytes / 16;
        for (int grid : {2048, 8192}) {
            float tr = 0, tw = 0, tc = 0;
            const int reps = 10;
            for (int r = 0; r < reps; ++r) {
                float ms;
                hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, flush, big / 16);      // push everything out of L2 / MALL
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, b); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); tr += ms;
                hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, flush, big / 16);
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); tw += ms;
                hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, flush, big / 16);
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); tc += ms;
            }
            printf("%4zu MB grid %5d: read %.1f us (%.2f TB/s) | write %.1f us (%.2f TB/s) | copy %.1f us (%.2f TB/s r+w)\n", bytes >> 20, grid,
                   tr / reps * 1e3, bytes / (tr / reps * 1e-3) / 1e12, tw / reps * 1e3, bytes / (tw / reps * 1e-3) / 1e12, tc / reps * 1e3,
                   2.0 * bytes / (tc / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}
