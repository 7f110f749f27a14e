// Minimal C++ caller of the C ABI (include/sttm_hip.h): no Python, no torch -- plain HIP allocations and the library.
// Build (see examples/build_demo.sh):  hipcc --offload-arch=gfx950 -Iinclude examples/c_abi_demo.cpp -Lsttm_amd/lib -lsttm_hip
//
// Merges one synthetic clip (T frames of H x W tokens with C channels: a few "scenes" of smooth random fields plus noise, so
// that both the spatial and the temporal stage have something to merge) and prints the counters.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sttm_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 32, H = 14, W = 14, C = argc > 2 ? atoi(argv[2]) : 256;
    const int N = T * H * W;
    printf("libsttm_hip ABI version %d\n", sttm_abi_version());

    // host-side synthetic clip, channels-last [T, H, W, C]
    std::vector<float> x((size_t)N * C);
    uint32_t rng = 12345u;
    auto uni = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) * (1.0f / 16777216.0f) - 0.5f; };
    std::vector<float> base((size_t)H * W * C);
    for (int t = 0; t < T; ++t) {
        if (t % 8 == 0)                                                      // a new "scene": smooth field, 2 x 2 blocks alike
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx)
                    for (int c = 0; c < C; ++c) {
                        if ((y & 1) == 0 && (xx & 1) == 0) base[((size_t)y * W + xx) * C + c] = uni();
                        else base[((size_t)y * W + xx) * C + c] = base[((size_t)(y & ~1) * W + (xx & ~1)) * C + c] + 0.05f * uni();
                    }
        for (size_t i = 0; i < (size_t)H * W * C; ++i) x[(size_t)t * H * W * C + i] = base[i] + 0.02f * uni();
    }

    void *dx, *ws, *feat, *npatch, *tlbr, *counts;
    const size_t ws_bytes = sttm_quadtree_workspace_bytes(T, H, W, C, STTM_F32, /*root_level=*/1);
    if (ws_bytes == 0) { fprintf(stderr, "workspace: %s\n", sttm_last_error()); return 2; }
    HIP_OK(hipMalloc(&dx, x.size() * 4));
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMalloc(&feat, (size_t)N * C * 4));
    HIP_OK(hipMalloc(&npatch, (size_t)N * 4));
    HIP_OK(hipMalloc(&tlbr, (size_t)N * 5 * 4));
    HIP_OK(hipMalloc(&counts, STTM_CNT_SLOTS * 4));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    // logical [T, C, H, W] with element strides (t, c, h, w) = (H*W*C, 1, W*C, C): the channels-last view
    const int rc = sttm_quadtree_merge(dx, (int64_t)H * W * C, 1, (int64_t)W * C, C, T, C, H, W, STTM_F32,
                                       /*threshold=*/0.85f, /*temporal_thresh=*/0.55f, /*root_level=*/1, /*weighted_avg=*/0,
                                       /*head_dim=*/0, /*slow_ver=*/0, ws, ws_bytes, feat, (int32_t*)npatch, (int32_t*)tlbr,
                                       (int32_t*)counts, stream);
    if (rc != STTM_OK) { fprintf(stderr, "sttm_quadtree_merge: %d %s\n", rc, sttm_last_error()); return 3; }
    int32_t h_counts[STTM_CNT_SLOTS];
    HIP_OK(hipMemcpyAsync(h_counts, counts, sizeof(h_counts), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    const int n_out = h_counts[STTM_CNT_OUT];
    std::vector<int32_t> h_np(n_out);
    HIP_OK(hipMemcpy(h_np.data(), npatch, (size_t)n_out * 4, hipMemcpyDeviceToHost));
    long patches = 0;
    for (int v : h_np) patches += v;
    printf("tokens %d -> merged %d (%.1f %%); spatial nodes %d, candidate pairs %d, kept pairs %d, label iterations %d, overflow %d\n",
           N, n_out, 100.0 * n_out / N, h_counts[STTM_CNT_NODES], h_counts[STTM_CNT_CANDIDATES], h_counts[STTM_CNT_EDGES],
           h_counts[STTM_CNT_ITERS], h_counts[STTM_CNT_OVERFLOW]);
    printf("sum of num_patches = %ld (must be %d)\n", patches, N);
    if (!(patches == N && h_counts[STTM_CNT_OVERFLOW] == 0 && n_out > 0 && n_out < N)) return 4;

    // The same merge through the argument block (ABI v5) with N' read from pinned host memory while the feature kernel may still
    // be running: no stream synchronisation between the call and the moment the caller knows how many rows it got.
    int32_t* counts_host = nullptr;
    uint64_t* early_host = nullptr;
    HIP_OK(hipHostMalloc((void**)&counts_host, STTM_CNT_SLOTS * sizeof(int32_t), hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&early_host, STTM_EARLY_SLOTS * sizeof(uint64_t), hipHostMallocDefault));
    for (int i = 0; i < STTM_CNT_SLOTS; ++i) counts_host[i] = 0;
    for (int i = 0; i < STTM_EARLY_SLOTS; ++i) early_host[i] = 0;
    sttm_merge_args g = {};          // (flags = 0: defaults)
    g.x = dx; g.stride_t = (int64_t)H * W * C; g.stride_c = 1; g.stride_h = (int64_t)W * C; g.stride_w = C;
    g.T = T; g.C = C; g.H = H; g.W = W; g.dtype = STTM_F32;
    g.threshold = 0.85f; g.temporal_thresh = 0.55f; g.root_level = 1;
    g.workspace = ws; g.workspace_bytes = ws_bytes;
    g.feat_out = feat; g.npatch_out = (int32_t*)npatch; g.tlbr_out = (int32_t*)tlbr; g.counts = (int32_t*)counts;
    g.counts_host = counts_host; g.early_host = early_host; g.seq = 1; g.stream = stream;
    const int rc2 = sttm_quadtree_merge_packed(&g);
    if (rc2 != STTM_OK) { fprintf(stderr, "sttm_quadtree_merge_packed: %d %s\n", rc2, sttm_last_error()); return 5; }
    int32_t early[2] = {0, 0};
    const int rc3 = sttm_wait_counts_early(counts_host, early_host, g.n_early, g.seq, 2000000, early);
    if (rc3 != STTM_OK) { fprintf(stderr, "sttm_wait_counts_early: %d\n", rc3); return 5; }
    printf("argument-block call: N' = %d from %d column words before any stream synchronisation (first call: %d)\n", early[0], g.n_early, n_out);
    HIP_OK(hipStreamSynchronize(stream));
    return (early[0] == n_out && early[1] == 0) ? 0 : 6;
}
