// K1-shaped load floor probe for MI355X (round 5, review item 1): how fast can the spatial kernel's ACCESS SHAPE stream the
// headline input -- 128 frames x 196 token rows of 4 KB, one 256-thread workgroup per (frame, root cell) with sixteen 16-byte
// loads per lane in flight -- when the kernel does nothing else, and which variation of the shape moves that number.
//
// Every kernel reads one 102.8 MB video ([T][196][1024] fp32) out of a pool of 8 (822 MB: past the 256 MB Infinity Cache, as
// in bench.py), sums what it loaded and stores nothing unless TAIL says so.  Variants (template parameter V):
//    0  lin16      : workgroup b reads rows 16b .. 16b+15 (64 KB contiguous), 1568 workgroups -- the streaming reference
//    1  k1         : K1's own shape: b -> (t = b / 16, root cell = b % 16), the root cell's leaf rows incl. the alias rows of
//                    the cells that own the lone first row / column (14 -> 7 -> 4: 7 of 16 root cells), 2048 workgroups
//    2  k1_noalias : the same without alias rows (196 loads per frame instead of 256)
//    3  k1_rcmajor : b -> (root cell = b / T, t = b % T): a root cell's frames are neighbours in launch order
//    4  k1_xcd     : every XCD (b % 8) walks its own contiguous range of 16 frames
//    5  k1_halves  : eight rows requested, consumed, then the other eight
//    6  k1_nt      : nontemporal loads
//    7  persist    : 1024 workgroups (4 per CU), each takes items b and b + 1024 one after the other
//    8  persist_pf : the same with the second item's sixteen loads requested before the first is consumed (32 packs live)
//    9  k1_pair    : 1024 workgroups, each loads TWO root cells (32 loads per lane in flight)
//   10  k1_sc1     : loads with sc1 (bypass the CU's vector L1)
// OCC: workgroups per CU the kernel is limited to through its dynamic LDS size (8 = no limit; K1 itself runs at 4).
// TAIL: microseconds every workgroup spends after its loads with nothing in flight (K1: ~2.6 us of decisions + emission), after
//       which it stores one 4 KB row (8 MB per launch; K1 stores ~14 MB).
// Build: hipcc --offload-arch=gfx950 -O3 -o k1_probe k1_probe.hip ; run plain (HIP-event times) or under rocprofv3 --kernel-trace --stats.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int T = 128, HW = 196, GW = 14, ROWB = 4096, R = 16, NBUF = 8;
constexpr size_t FRAME_B = (size_t)HW * ROWB, VIDEO_B = (size_t)T * FRAME_B;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Tab { int row[R][16]; };        // row index inside the frame of leaf slot s of root cell rc; -1 = no load

__device__ __forceinline__ f4 ld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int aux) {
    u4 v;
    if (aux == 2) v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 2);
    else if (aux == 16) v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
    else v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    f4 r; __builtin_memcpy(&r, &v, 16); return r;
}

__device__ __forceinline__ void item_of(int V, int b, int& t, int& rc) {
    if (V == 3) { rc = b / T; t = b % T; }
    else if (V == 4) { const int x = b & 7, j = b >> 3; t = x * (T / 8) + j / R; rc = j % R; }
    else { t = b / R; rc = b % R; }
}

template <int N, int AUX>
__device__ __forceinline__ void load_rows(const char* vid, const Tab& tab, int t, int rc, int s0, f4 (&p)[N], bool alias) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vid) + (size_t)t * FRAME_B, 0, -1, 0x00020000);
    const unsigned voff = threadIdx.x * 16u;
    int rows[N];
#pragma unroll
    for (int s = 0; s < N; ++s) rows[s] = tab.row[rc][s0 + s];      // one wide scalar load of the table row
#pragma unroll
    for (int s = 0; s < N; ++s) {
        int row = rows[s];
        const bool own = row >= 0;
        if (!own) row = alias ? -row - 1 : -1;
        row = __builtin_amdgcn_readfirstlane(row);
        if (row >= 0) p[s] = ld(rs, voff, (unsigned)row * ROWB, AUX);
        else p[s] = f4{0, 0, 0, 0};
    }
}

template <int N>
__device__ __forceinline__ f4 sum_rows(const f4 (&p)[N]) {
    f4 a = p[0];
#pragma unroll
    for (int s = 1; s < N; ++s) a += p[s];
    return a;
}

__device__ __forceinline__ void tail_and_store(f4 acc, int tail_us, f4* out, int b) {
    if (tail_us > 0) {
        const long long t0 = wall_clock64();             // 100 MHz
        while (wall_clock64() - t0 < (long long)tail_us * 100) __builtin_amdgcn_s_sleep(4);
        out[(size_t)b * 256 + threadIdx.x] = acc;
    } else if (acc.x == 1234567.f) out[threadIdx.x] = acc;
}

template <int V, int OCC, int TAIL>
__global__ void __launch_bounds__(256) k_probe(const char* __restrict__ vid, const Tab tab, f4* __restrict__ out) {
    extern __shared__ char lds[];
    const int b = blockIdx.x;
    if (threadIdx.x == 9999) lds[0] = 1;
    if constexpr (V == 0) {
        f4 p[16];
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vid), 0, -1, 0x00020000);
        const unsigned voff = threadIdx.x * 16u;
#pragma unroll
        for (int s = 0; s < 16; ++s) p[s] = ld(rs, voff, (unsigned)(b * 16 + s) * ROWB, 0);
        tail_and_store(sum_rows(p), TAIL, out, b);
    } else if constexpr (V == 5) {
        int t, rc; item_of(V, b, t, rc);
        f4 p[8], q[8];
        load_rows<8, 0>(vid, tab, t, rc, 0, p, true);
        f4 a = sum_rows(p);
        load_rows<8, 0>(vid, tab, t, rc, 8, q, true);
        a += sum_rows(q);
        tail_and_store(a, TAIL, out, b);
    } else if constexpr (V == 7 || V == 8) {
        int t, rc; item_of(1, b, t, rc);
        f4 p[16], q[16];
        load_rows<16, 0>(vid, tab, t, rc, 0, p, true);
        int t2, rc2; item_of(1, b + 1024, t2, rc2);
        f4 a;
        if constexpr (V == 8) {
            load_rows<16, 0>(vid, tab, t2, rc2, 0, q, true);
            a = sum_rows(p);
            tail_and_store(a, TAIL, out, b);
            a = sum_rows(q);
        } else {
            a = sum_rows(p);
            tail_and_store(a, TAIL, out, b);
            load_rows<16, 0>(vid, tab, t2, rc2, 0, q, true);
            a = sum_rows(q);
        }
        tail_and_store(a, TAIL, out, b + 1024);
    } else if constexpr (V == 9) {
        int t, rc; item_of(1, 2 * b, t, rc);
        f4 p[16], q[16];
        load_rows<16, 0>(vid, tab, t, rc, 0, p, true);
        load_rows<16, 0>(vid, tab, t, rc + 1, 0, q, true);
        tail_and_store(sum_rows(p) + sum_rows(q), TAIL, out, b);
    } else {
        int t, rc; item_of(V, b, t, rc);
        f4 p[16];
        constexpr int AUX = V == 6 ? 2 : (V == 10 ? 16 : 0);
        load_rows<16, AUX>(vid, tab, t, rc, 0, p, V != 2);
        tail_and_store(sum_rows(p), TAIL, out, b);
    }
}

// leaf rows of the 3-level tree 14 -> 7 -> 4 (odd level 7: the first cell of an axis stands alone), alias rows as -(row)-1
static int child_start(int i, int n_child) { return (n_child & 1) ? (i == 0 ? 0 : 2 * i - 1) : 2 * i; }
static int child_count(int i, int n_child) { return ((n_child & 1) && i == 0) ? 1 : 2; }
static Tab make_tab() {
    Tab tb;
    for (int I = 0; I < 4; ++I)
        for (int J = 0; J < 4; ++J) {
            const int rc = I * 4 + J;
            for (int k = 0; k < 4; ++k)
                for (int q = 0; q < 4; ++q) {
                    const int dy = k >> 1, dx = k & 1, ey = q >> 1, ex = q & 1;
                    const int nmy = child_count(I, 7), nmx = child_count(J, 7);
                    const bool mid_ok = dy < nmy && dx < nmx;
                    int row;
                    if (mid_ok) {
                        const int my = child_start(I, 7) + dy, mx = child_start(J, 7) + dx;      // level-7 cell
                        const int y = child_start(my, 14) + ey, x = child_start(mx, 14) + ex;    // level 14 is even: always 2 x 2
                        row = y * GW + x;
                    } else {
                        row = -((ey * GW + ex) + 1);                                             // leaf of mid (0, 0): the alias of a missing mid
                    }
                    tb.row[rc][4 * k + q] = row;
                }
        }
    return tb;
}

struct Result { const char* name; int grid; double us_avg, us_min; double mb; };
static std::vector<Result> g_res;

template <int V, int OCC, int TAIL>
static void run(const char* name, char* const* bufs, const Tab& tab, f4* out, int reps) {
    const int grid = V == 0 ? (T * HW) / 16 : ((V == 7 || V == 8 || V == 9) ? 1024 : T * R);
    const size_t lds = OCC >= 8 ? 0 : (size_t)(160 * 1024 / OCC) - 1024;
    if (lds) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<V, OCC, TAIL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double tot = 0, mn = 1e30;
    for (int i = 0; i < reps + 3; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<V, OCC, TAIL>), dim3(grid), dim3(256), lds, 0, bufs[i % NBUF], tab, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) { tot += ms * 1e3; mn = std::min(mn, (double)ms * 1e3); }
    }
    const double mb = (V == 2 ? T * 196.0 : (V == 0 ? T * 196.0 : T * 256.0)) * ROWB / 1e6;
    g_res.push_back({name, grid, tot / reps, mn, mb});
    printf("%-44s grid %5d  avg %7.2f us  min %7.2f us  requested %6.1f MB  -> %5.2f TB/s of unique 102.8 MB (min)\n", name, grid, tot / reps, mn, mb,
           102.76 / mn);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 24;
    char* bufs[NBUF];
    for (int i = 0; i < NBUF; ++i) {
        CHECK(hipMalloc(&bufs[i], VIDEO_B));
        CHECK(hipMemset(bufs[i], 0x11 + i, VIDEO_B));
    }
    f4* out;
    CHECK(hipMalloc(&out, (size_t)2048 * 4096));
    const Tab tab = make_tab();
    CHECK(hipDeviceSynchronize());
#define RUN(V, OCC, TAIL, NAME) run<V, OCC, TAIL>(NAME " occ" #OCC " tail" #TAIL, bufs, tab, out, reps)
    RUN(0, 8, 0, "lin16");
    RUN(0, 4, 0, "lin16");
    RUN(1, 8, 0, "k1");
    RUN(1, 4, 0, "k1");
    RUN(1, 2, 0, "k1");
    RUN(2, 4, 0, "k1_noalias");
    RUN(3, 4, 0, "k1_rcmajor");
    RUN(4, 4, 0, "k1_xcd");
    RUN(5, 4, 0, "k1_halves");
    RUN(6, 4, 0, "k1_nt");
    RUN(10, 4, 0, "k1_sc1");
    RUN(7, 4, 0, "persist");
    RUN(8, 4, 0, "persist_pf");
    RUN(9, 4, 0, "k1_pair");
    // with K1's tail: nothing in flight for 2 / 4 us per workgroup, then one 4 KB row stored
    RUN(0, 4, 2, "lin16");
    RUN(1, 4, 2, "k1");
    RUN(1, 4, 4, "k1");
    RUN(1, 8, 2, "k1");
    RUN(4, 4, 2, "k1_xcd");
    RUN(7, 4, 2, "persist");
    RUN(8, 4, 2, "persist_pf");
    RUN(9, 4, 2, "k1_pair");
    RUN(6, 4, 2, "k1_nt");
    return 0;
}
