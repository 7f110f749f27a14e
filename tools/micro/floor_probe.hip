// Floors for the row-moving kernels of the temporal stage on MI355X: how fast can 11.2 k rows of 4 KB be gathered out of a
// 103 MB matrix (just read by another kernel, i.e. Infinity-Cache resident) into a dense 46 MB output when the row indices are
// KNOWN up front (the group-mean kernel has to find them first), and how fast can the pair kernel's rows be read.
// Build: hipcc --offload-arch=gfx950 -O3 -o floor_probe floor_probe.hip ; run under rocprofv3 --kernel-trace --stats
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ROW_F4 = 256;   // 4 KB rows

// warm: read the whole source (what the pair kernel leaves behind)
__global__ void k_warm(const f4* __restrict__ x, size_t n, f4* sink) {
    f4 acc = {0, 0, 0, 0};
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 + 3 < n) { const f4 a = x[i0], b = x[i0 + 1], c = x[i0 + 2], d = x[i0 + 3]; acc = a + b + c + d; }
    if (acc.x == 12345.678f) sink[0] = acc;
}
// one wave per RPW rows, all 4 * RPW loads of 16 B issued before the first store
template <int RPW>
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ x, const int* __restrict__ idx, int n_rows, f4* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    f4 v[RPW][4];
    int src[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) { const int o = wave * RPW + r; src[r] = o < n_rows ? idx[o] : -1; }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (src[r] >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[r][k] = x[(size_t)src[r] * ROW_F4 + k * 64 + lane];
        }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (src[r] >= 0) {
            const int o = wave * RPW + r;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[(size_t)o * ROW_F4 + k * 64 + lane] = v[r][k];
        }
}
// same, nontemporal stores
template <int RPW>
__global__ void __launch_bounds__(256) k_gather_nt(const f4* __restrict__ x, const int* __restrict__ idx, int n_rows, f4* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    f4 v[RPW][4];
    int src[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) { const int o = wave * RPW + r; src[r] = o < n_rows ? idx[o] : -1; }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (src[r] >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[r][k] = x[(size_t)src[r] * ROW_F4 + k * 64 + lane];
        }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (src[r] >= 0) {
            const int o = wave * RPW + r;
#pragma unroll
            for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[r][k], &y[(size_t)o * ROW_F4 + k * 64 + lane]);
        }
}
// persistent form: grid of `gridDim.x` workgroups, every wave walks rows wave, wave + nwaves, ... with the next row's loads
// issued before the current row's stores
__global__ void __launch_bounds__(256) k_gather_persist(const f4* __restrict__ x, const int* __restrict__ idx, int n_rows, f4* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    int o = wave;
    if (o >= n_rows) return;
    f4 cur[4];
    int s = idx[o];
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = x[(size_t)s * ROW_F4 + k * 64 + lane];
    while (true) {
        const int on = o + nw;
        f4 nxt[4];
        const bool more = on < n_rows;
        if (more) {
            const int sn = idx[on];
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[k] = x[(size_t)sn * ROW_F4 + k * 64 + lane];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) y[(size_t)o * ROW_F4 + k * 64 + lane] = cur[k];
        if (!more) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        o = on;
    }
}
// two-stage form like the group-mean kernel: the wave first reads 196 int32 of metadata of "its frame", then moves its row
template <int RPW>
__global__ void __launch_bounds__(256) k_gather_2stage(const f4* __restrict__ x, const int* __restrict__ idx, const int* __restrict__ metadata,
                                                       int n_rows, f4* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int m = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) m += metadata[((wave * 4 + b) * 64 + lane) & 0xffff];
    m = __builtin_amdgcn_readfirstlane(m) & 0;              // a real dependency, value 0
    f4 v[RPW][4];
    int src[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) { const int o = wave * RPW + r; src[r] = o < n_rows ? idx[o + m] : -1; }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (src[r] >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[r][k] = x[(size_t)src[r] * ROW_F4 + k * 64 + lane];
        }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (src[r] >= 0) {
            const int o = wave * RPW + r;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[(size_t)o * ROW_F4 + k * 64 + lane] = v[r][k];
        }
}
// pair-kernel floor: workgroup = 4 waves, wave w reads NC "candidates" = 2 rows each with all loads in flight (LOADS of 16 B per
// lane and pass), sums, keeps one value
template <int NC, int LOADS>
__global__ void __launch_bounds__(256) k_pairread(const f4* __restrict__ x, const int* __restrict__ rows, int n_wg, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int c = 0; c < NC; ++c) {
        const int ra = rows[(blockIdx.x * 4 + wave) * 2 * NC + 2 * c], rb = rows[(blockIdx.x * 4 + wave) * 2 * NC + 2 * c + 1];
        for (int k0 = 0; k0 < 4; k0 += LOADS) {
            f4 a[LOADS], b[LOADS];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) { a[k] = x[(size_t)ra * ROW_F4 + (k0 + k) * 64 + lane]; b[k] = x[(size_t)rb * ROW_F4 + (k0 + k) * 64 + lane]; }
#pragma unroll
            for (int k = 0; k < LOADS; ++k) acc += a[k].x * b[k].x + a[k].y * b[k].y + a[k].z * b[k].z + a[k].w * b[k].w;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) out[blockIdx.x * 4 + wave] = acc;
}

// the group-mean kernel's singleton path, minimal: workgroup (t, s) of 4 waves, every wave reads the frame's metadata (NMETA arrays x
// 4 chunks of 64 slots), ranks the survivors with ballots, takes the rank congruent to its id, copies that row.  S workgroups per frame.
template <int NMETA, bool TLBR>
__global__ void __launch_bounds__(256) k_gm_like(const f4* __restrict__ x, const int* __restrict__ gcnt, const int* __restrict__ meta,
                                                 const int* __restrict__ geo, const int* __restrict__ frame_cnt, int S, int HW,
                                                 f4* __restrict__ y, int* __restrict__ tlbr) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x / S, s = blockIdx.x - t * S;
    const int stride = S * 4, me = s * 4 + wave;
    int row0 = 0;
    for (int f = lane; f < t; f += 64) row0 += frame_cnt[f];
    int cnt[4], mt[4], gg[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int p = b * 64 + lane, pc = p < HW ? p : HW - 1;
        cnt[b] = gcnt[t * HW + pc];
        mt[b] = NMETA >= 2 ? meta[t * HW + pc] : 0;
        gg[b] = NMETA >= 3 ? geo[pc] : 0;
        if (p >= HW) cnt[b] = 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) row0 += __shfl_xor(row0, d, 64);
    int j0 = 0, sel_p = -1, sel_j = 0, extra = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const unsigned long long m = __ballot(cnt[b] > 0);
        const int j = j0 + __popcll(m & ((1ull << lane) - 1ull));
        const unsigned long long h = __ballot(cnt[b] > 0 && (j & (stride - 1)) == me);
        if (h) { const int l = __ffsll((long long)h) - 1; sel_p = b * 64 + l; sel_j = j0 + __popcll(m & ((1ull << l) - 1ull));
                 extra = __builtin_amdgcn_readlane(mt[b], l) + __builtin_amdgcn_readlane(gg[b], l); }
        j0 += __popcll(m);
    }
    if (sel_p < 0) return;
    const int src = t * HW + sel_p, o = row0 + sel_j;
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x[(size_t)src * ROW_F4 + k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[(size_t)o * ROW_F4 + k * 64 + lane] = v[k];
    if (TLBR && lane == 0) { int* q = tlbr + (size_t)o * 6; q[0] = t; q[1] = sel_p; q[2] = extra; q[3] = 1; q[4] = 2; q[5] = 3; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int T = 128, HW = 196, NROWS = T * HW, NOUT = 11200;
    const size_t src_bytes = (size_t)NROWS * 4096, dst_bytes = (size_t)NOUT * 4096;
    f4 *src, *dst, *flush, *sink; int *idx, *meta, *prow; float* pout;
    CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&dst, dst_bytes)); CK(hipMalloc(&flush, 1ull << 30)); CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&idx, NROWS * 4)); CK(hipMalloc(&meta, 65536 * 4)); CK(hipMemset(meta, 0, 65536 * 4));
    CK(hipMemset(src, 1, src_bytes));
    std::vector<int> h(NOUT);
    srand(1);
    { std::vector<int> all(NROWS); for (int i = 0; i < NROWS; ++i) all[i] = i;
      for (int i = 0; i < NOUT; ++i) { const int j = i + rand() % (NROWS - i); std::swap(all[i], all[j]); }
      std::vector<int> pick(all.begin(), all.begin() + NOUT); std::sort(pick.begin(), pick.end()); h = pick; }
    CK(hipMemcpy(idx, h.data(), NOUT * 4, hipMemcpyHostToDevice));
    // pair rows: 2032 workgroups, 4 waves, 2 candidates of 2 rows in neighbouring frames
    const int NWG = 2032, NC = 2;
    std::vector<int> pr((size_t)NWG * 4 * 2 * NC);
    for (int w = 0; w < NWG; ++w) { const int t = w % 127, r = w / 127;
        for (int k = 0; k < 4 * NC; ++k) { const int p = (r * 12 + rand() % 12) % HW; pr[((size_t)w * 4 * NC + k) * 2] = t * HW + p; pr[((size_t)w * 4 * NC + k) * 2 + 1] = (t + 1) * HW + (p + rand() % 3) % HW; } }
    CK(hipMalloc(&prow, pr.size() * 4)); CK(hipMemcpy(prow, pr.data(), pr.size() * 4, hipMemcpyHostToDevice)); CK(hipMalloc(&pout, NWG * 4 * 4));
    // metadata for k_gm_like: ~87 survivors per frame
    std::vector<int> hg((size_t)T * HW, 0), hm((size_t)T * HW, 7), hgeo(HW, 3), hfc(T, 0);
    for (int t = 0; t < T; ++t) { std::vector<int> sl(HW); for (int i = 0; i < HW; ++i) sl[i] = i;
        const int ns = 87 + (t & 1);
        for (int i = 0; i < ns; ++i) { const int j = i + rand() % (HW - i); std::swap(sl[i], sl[j]); hg[(size_t)t * HW + sl[i]] = 1; }
        hfc[t] = ns; }
    int *dgc, *dmt, *dgeo, *dfc, *dtl;
    CK(hipMalloc(&dgc, hg.size() * 4)); CK(hipMalloc(&dmt, hm.size() * 4)); CK(hipMalloc(&dgeo, HW * 4)); CK(hipMalloc(&dfc, T * 4));
    CK(hipMalloc(&dtl, (size_t)NROWS * 6 * 4));
    CK(hipMemcpy(dgc, hg.data(), hg.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dmt, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgeo, hgeo.data(), HW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dfc, hfc.data(), T * 4, hipMemcpyHostToDevice));
    const size_t nsrc4 = src_bytes / 16;
    for (int rep = 0; rep < 20; ++rep) {
#define WARM() hipLaunchKernelGGL(k_warm, dim3((unsigned)((nsrc4 / 4 + 255) / 256)), dim3(256), 0, 0, src, nsrc4, sink)
        WARM(); hipLaunchKernelGGL((k_gather<1>), dim3((NOUT + 3) / 4), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gather<2>), dim3((NOUT + 7) / 8), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gather<4>), dim3((NOUT + 15) / 16), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gather_nt<1>), dim3((NOUT + 3) / 4), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gather_nt<2>), dim3((NOUT + 7) / 8), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL(k_gather_persist, dim3(2048), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL(k_gather_persist, dim3(1024), dim3(256), 0, 0, src, idx, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gather_2stage<1>), dim3((NOUT + 3) / 4), dim3(256), 0, 0, src, idx, meta, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gather_2stage<2>), dim3((NOUT + 7) / 8), dim3(256), 0, 0, src, idx, meta, NOUT, dst);
        WARM(); hipLaunchKernelGGL((k_gm_like<1, false>), dim3(T * 32), dim3(256), 0, 0, src, dgc, dmt, dgeo, dfc, 32, HW, dst, dtl);
        WARM(); hipLaunchKernelGGL((k_gm_like<3, false>), dim3(T * 32), dim3(256), 0, 0, src, dgc, dmt, dgeo, dfc, 32, HW, dst, dtl);
        WARM(); hipLaunchKernelGGL((k_gm_like<3, true>), dim3(T * 32), dim3(256), 0, 0, src, dgc, dmt, dgeo, dfc, 32, HW, dst, dtl);
        WARM(); hipLaunchKernelGGL((k_gm_like<3, true>), dim3(T * 16), dim3(256), 0, 0, src, dgc, dmt, dgeo, dfc, 16, HW, dst, dtl);
        WARM(); hipLaunchKernelGGL((k_pairread<2, 2>), dim3(NWG), dim3(256), 0, 0, src, prow, NWG, pout);
        WARM(); hipLaunchKernelGGL((k_pairread<2, 4>), dim3(NWG), dim3(256), 0, 0, src, prow, NWG, pout);
        WARM(); hipLaunchKernelGGL((k_pairread<2, 1>), dim3(NWG), dim3(256), 0, 0, src, prow, NWG, pout);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
