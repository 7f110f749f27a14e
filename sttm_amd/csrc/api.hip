// C-ABI entry points of libsttm_hip.so (declared in include/sttm_hip.h).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sched.h>

#include "sttm_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Tuning / test switches: defaults from STTM_<KEY> environment variables, read ONCE; sttm_configure overrides them.
// None of them changes results.
struct Config {
    int pairs_seg, pairs_nt, pairs_var, k1_var, k1_split, no_dense, gm_split, label_nt, vec16, vec32, fold_kb, fold_labels, no_fuse, force_gmem_labels, tome_split, tome_flat,
        batch_streams, batch_sub, col_walk, col_frames, col_cap, col_pb, col_abl, pair_vec;
};
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
Config& config() {
    static Config c = [] {
        Config d;
        d.pairs_seg = env_int("STTM_PAIRS_SEG", 0);
        d.pairs_nt = env_int("STTM_PAIRS_NT", 0);
        d.pairs_var = env_int("STTM_PAIRS_VAR", 0);
        d.k1_var = env_int("STTM_K1_VAR", 0);
        d.k1_split = env_int("STTM_K1_SPLIT", 0);
        d.no_dense = env_int("STTM_NO_DENSE", 0);
        d.gm_split = env_int("STTM_GM_SPLIT", 0);
        d.label_nt = env_int("STTM_LABEL_NT", 0);
        d.vec16 = env_int("STTM_VEC16", 0);
        d.vec32 = env_int("STTM_VEC32", 0);
        d.fold_kb = env_int("STTM_FOLD_KB", 64);
        d.fold_labels = env_int("STTM_FOLD_LABELS", 0);
        d.no_fuse = env_int("STTM_NO_FUSE", env_int("STTM_NO_FUSE_LABELS", 0));
        d.force_gmem_labels = env_int("STTM_FORCE_GMEM_LABELS", 0);
        d.tome_split = env_int("STTM_TOME_SPLIT", 2);
        d.tome_flat = env_int("STTM_TOME_FLAT", 1);
        d.batch_streams = env_int("STTM_BATCH_STREAMS", 3);
        d.batch_sub = env_int("STTM_BATCH_SUB", 8);
        // the column-walk spatial stage (spatial_col.inc): 0 = never (default: measured slower than the spatial + pair kernels although it
        // moves 17 % fewer bytes -- DESIGN.md 4.5), 1 = launch sets of several videos, 2 = also a single video (tests / A-B)
        d.col_walk = env_int("STTM_COL_WALK", 0);
        d.col_frames = env_int("STTM_COL_FRAMES", 8);
        d.col_cap = env_int("STTM_COL_CAP", 0);            // 0 = what the LDS budget allows
        d.col_pb = env_int("STTM_COL_PB", 0);
#ifdef STTM_DEV
        d.col_abl = env_int("STTM_COL_ABL", 0);          // development build: ablation bits of the column walk (outputs invalid)
#else
        d.col_abl = 0;
#endif
        d.pair_vec = env_int("STTM_PAIR_VEC", 0);          // 0 = the column walk's width where it could run (see merge_group); -1 = the row kernels' width (rounds 1-5)
        return d;
    }();
    return c;
}

#ifdef STTM_DEV
sttm::DevHooks g_dev = {};
#endif

inline int ceil_half(int v) { return (v + 1) / 2; }

// The spatial kernels read token rows through a buffer descriptor per frame with 32-bit row offsets (quadtree_spatial.inc):
// a frame must span < 2 GiB and the strides must not be negative (any channels-last view of a real tensor qualifies).
int check_frame_addressing(int H, int W, int C, int eb, int64_t stride_h, int64_t stride_w, int64_t stride_t) {
    if (stride_t < 0 || stride_h < 0 || stride_w < 0)
        return fail(STTM_ERR_UNSUPPORTED, "negative strides are not supported (make the input contiguous in [T, H, W, C])");
    const long double ext = ((long double)(H - 1) * stride_h + (long double)(W - 1) * stride_w + C) * eb;
    if (ext >= 2147483648.0L)
        return fail(STTM_ERR_UNSUPPORTED, "one frame of the input view spans %.1f GiB; the spatial kernel addresses a frame with 32-bit offsets "
                    "(make the input contiguous in [T, H, W, C])", (double)(ext / 1073741824.0L));
    return STTM_OK;
}

// Level list of quadtree_builder.py:101-117: halve (ceil) until EITHER side is 2, then build pyramid levels
// until the WIDTH equals the width of entry `root_level` (negative indices count from the fine end).
int build_dims(int H, int W, int root_level, sttm::LevelDims* out) {
    if (H < 1 || W < 1 || (H == 1 && W == 1)) return fail(STTM_ERR_ARG, "degenerate token grid %dx%d", H, W);
    int hs[64], ws[64], n = 1;
    hs[0] = H; ws[0] = W;                       // fine -> coarse here
    int h = H, w = W;
    while (h != 2 && w != 2) {
        w = ceil_half(w); h = ceil_half(h);
        if (n >= 63) return fail(STTM_ERR_ARG, "token grid %dx%d never reaches a side of 2", H, W);
        hs[n] = h; ws[n] = w; ++n;
    }
    // python index into the coarse->fine list of length n
    int idx = root_level < 0 ? root_level + n : root_level;
    if (idx < 0 || idx >= n) return fail(STTM_ERR_INDEX, "root_level %d out of range for %d levels", root_level, n);
    const int target_w = ws[n - 1 - idx];
    int dh[64], dw[64], m = 1;
    dh[0] = H; dw[0] = W;                       // fine -> coarse
    while (dw[m - 1] != target_w) {
        if (m >= 63) return fail(STTM_ERR_ARG, "pyramid never reaches the requested root level");
        dh[m] = ceil_half(dh[m - 1]); dw[m] = ceil_half(dw[m - 1]); ++m;
    }
    if (m > sttm::kMaxLevels)
        return fail(STTM_ERR_UNSUPPORTED, "%d pyramid levels (grid %dx%d, root_level %d); the device path supports <= %d",
                    m, H, W, root_level, sttm::kMaxLevels);
    memset(out, 0, sizeof(*out));
    out->n_level = m;
    for (int l = 0; l < m; ++l) { out->h[l] = dh[m - 1 - l]; out->w[l] = dw[m - 1 - l]; }
    return m;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int elem_bytes(int dtype) { return dtype == STTM_F32 ? 4 : 2; }

struct Carve {
    char* base; size_t off;
    template <typename P> P* take(size_t bytes) {
        P* p = reinterpret_cast<P*>(base + off);
        off = align_up(off + bytes, 256);
        return p;
    }
};

struct Plan {
    sttm::LevelDims dims;
    int R, rc_stride, ecap, N, max_slots, max_area;
    size_t bytes;
};

// leaf extent of root cell (I, J): follow first / last children down to the leaf level
void root_extent_host(const sttm::LevelDims& g, int I, int J, int* ah, int* aw) {
    int lo_i = I, hi_i = I, lo_j = J, hi_j = J;
    for (int m = 0; m < g.n_level - 1; ++m) {
        lo_i = sttm::child_start(lo_i, g.h[m + 1]);
        hi_i = sttm::child_start(hi_i, g.h[m + 1]) + sttm::child_count(hi_i, g.h[m + 1]) - 1;
        lo_j = sttm::child_start(lo_j, g.w[m + 1]);
        hi_j = sttm::child_start(hi_j, g.w[m + 1]) + sttm::child_count(hi_j, g.w[m + 1]) - 1;
    }
    *ah = hi_i - lo_i + 1;
    *aw = hi_j - lo_j + 1;
}

struct Buffers {
    char* S; uint32_t* meta; double* inrm; int* rc_list;
    int32_t *edges; float* edge_sim; int32_t *edge_cnt, *cand_cnt; unsigned long long* col_mask; int32_t *col_arrive, *frame_cnt, *bar, *colscratch;
    int32_t *lab_row, *gcnt; uint32_t* cgeo;
    char* tops;               // trees of 4 and more levels, the split spatial stage: [T][blocks][C] pooled tops of the 3-level blocks,
                              // then [T * R][upper nodes][C] pooled features of the cells above them
};

size_t carve_all(const Plan& p, int T, int H, int W, int C, int dtype, char* base, Buffers* b) {
    Carve c{base, 0};
    const size_t N = (size_t)p.N;
    const size_t nfr = (size_t)(T > 1 ? T - 1 : 1) * p.R;
    Buffers tmp;
    Buffers& o = b ? *b : tmp;
    o.S = c.take<char>(N * C * elem_bytes(dtype));
    o.meta = c.take<uint32_t>(N * 4);
    o.inrm = c.take<double>(N * 8);
    o.rc_list = c.take<int>((size_t)T * p.R * p.rc_stride * 4);
    o.edges = c.take<int32_t>(nfr * p.ecap * 4);
    o.edge_sim = c.take<float>(nfr * p.ecap * 4);
    o.edge_cnt = c.take<int32_t>(nfr * 4);
    o.cand_cnt = c.take<int32_t>(nfr * 4);
    o.col_mask = c.take<unsigned long long>((size_t)p.R * 8);
    o.col_arrive = c.take<int32_t>((size_t)p.R * 4);
    o.frame_cnt = c.take<int32_t>((size_t)T * 4);
    o.bar = c.take<int32_t>(16);
    o.colscratch = c.take<int32_t>(sttm::colscratch_ints(T, H, W, p.R) * 4);
    o.lab_row = c.take<int32_t>(N * 4);
    o.gcnt = c.take<int32_t>(N * 4);
    o.cgeo = c.take<uint32_t>((size_t)H * W * 4);
    o.tops = c.take<char>(sttm::split_tops_bytes(T, p.dims, C, elem_bytes(dtype)) + sttm::split_ufeat_bytes(T, p.dims, C, elem_bytes(dtype)));
    return c.off;
}

int make_plan(int T, int H, int W, int C, int dtype, int root_level, Plan* p) {
    const int D = build_dims(H, W, root_level, &p->dims);
    if (D < 0) return D;
    p->R = p->dims.h[0] * p->dims.w[0];
    int max_area = 1;
    for (int I = 0; I < p->dims.h[0]; ++I)
        for (int J = 0; J < p->dims.w[0]; ++J) {
            int ah, aw;
            root_extent_host(p->dims, I, J, &ah, &aw);
            if (ah * aw > max_area) max_area = ah * aw;
        }
    p->max_area = max_area;
    p->rc_stride = 1 + max_area;
    p->ecap = 2 * max_area;
    p->N = T * H * W;
    p->max_slots = T * max_area;
    p->bytes = carve_all(*p, T, H, W, C, dtype, nullptr, nullptr);
    return D;
}

int pick_vec(int C, int dtype, const void* x, int64_t sT, int64_t sH, int64_t sW, int* nt, bool allow_wide16 = false) {
    const int eb = elem_bytes(dtype);
    const Config& cfg = config();
    if (allow_wide16 && dtype != STTM_F32) {
        // 32-byte packs for rows that need 5-8 waves with 16-byte packs (2048 < C <= 4096): four waves per root cell instead
        // of seven halve the per-wave reduction butterflies.  Measured on T=128 bf16: C=3584 75.1 -> 65.0 us; no gain at
        // C=8192 (16 -> 8 waves) and a loss at C=2048 (4 -> 2 waves), hence the window.  vec16 = 8 / 16 force a width.
        const int want = cfg.vec16;
        const int w8 = (C / 8 + 63) / 64;
        const bool window = C % 8 == 0 && w8 > 4 && w8 <= 8;
        if ((want == 16 || (want == 0 && window)) && C % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 32 == 0 &&
            (sT * eb) % 32 == 0 && (sH * eb) % 32 == 0 && (sW * eb) % 32 == 0 && C / 16 <= 512) {
            *nt = ((C / 16 + 63) / 64) * 64;
            return 16;
        }
    }
    if (allow_wide16 && dtype == STTM_F32) {
        // same idea for fp32 (32-byte packs); window mirrors the 16-bit one (5-8 waves with 16-byte packs: 1024 < C <= 2048)
        const int want32 = cfg.vec32;
        const int w4 = (C / 4 + 63) / 64;
        const bool window = C % 4 == 0 && w4 > 4 && w4 <= 8;
        if ((want32 == 8 || (want32 == 0 && window)) && C % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 32 == 0 &&
            (sT * eb) % 32 == 0 && (sH * eb) % 32 == 0 && (sW * eb) % 32 == 0 && C / 8 <= 512) {
            *nt = ((C / 8 + 63) / 64) * 64;
            return 8;
        }
    }
    // 16 bytes per lane for every dtype.  For 16-bit inputs the 8-wide pack became affordable (128 VGPRs) once the
    // dot products moved to v_dot2c_f32_bf16/f16; measured on T=128, C=3584 bf16: spatial 107 -> 84 us,
    // group mean 69 -> 51 us, pairs 42 -> 31 us versus 4-wide packs (vec16 = 4 restores them).
    int cands_f32[] = {4, 2, 1}, cands_16[] = {8, 4, 2};
    if (cfg.vec16 == 4) { cands_16[0] = 4; cands_16[1] = 8; }
    const int* cands = dtype == STTM_F32 ? cands_f32 : cands_16;
    for (int k = 0; k < 3; ++k) {
        const int v = cands[k];
        if (C % v) continue;
        const size_t ab = (size_t)v * eb;
        if (reinterpret_cast<uintptr_t>(x) % ab) continue;
        if ((sT * eb) % ab || (sH * eb) % ab || (sW * eb) % ab) continue;
        const int lanes = (C + v - 1) / v;
        if (lanes > 1024) continue;
        // prefer 16-byte lanes; for 16-bit types a narrower pack halves the VGPRs at equal coalescing
        // when the whole row still fits one workgroup
        *nt = ((lanes + 63) / 64) * 64;
        return v;
    }
    return 0;
}

// `sim_f32 >= thr_f32`  <=>  `sim >= lo` with lo = the midpoint below the fp32 threshold (exclusive when the tie would round
// down to the predecessor, i.e. when the threshold's mantissa is odd); returned as lo * |lo| for the division-free test.
double thr_lo_sq_of(float thr) {
    const float pred = nextafterf(thr, -INFINITY);
    double lo = 0.5 * ((double)pred + (double)thr);
    uint32_t bits; memcpy(&bits, &thr, 4);
    if (bits & 1u) lo = nextafter(lo, (double)INFINITY);
    return lo * fabs(lo);
}

// Octree level list (octree_utils.py:312-316): [2, ..., side]; the pyramid runs from sizes[root_level] down to side.
// Returns the number of pyramid levels (root first) or a negative error code.
int octree_levels(int side, int root_level, int* sides /*[kOctMaxLevels]*/) {
    if (side < 2) return STTM_ERR_ARG;
    int list[32], n = 0;
    int w = side;
    list[n++] = w;
    while (w != 2 && n < 32) { w = (w + 1) / 2; list[n++] = w; }        // fine -> coarse
    if (w != 2) return STTM_ERR_ARG;
    int idx = root_level < 0 ? n + root_level : root_level;             // index into the coarse -> fine list
    if (idx < 0 || idx >= n) return STTM_ERR_INDEX;
    const int L = n - idx;
    if (L > sttm::kOctMaxLevels) return STTM_ERR_UNSUPPORTED;
    for (int l = 0; l < L; ++l) sides[l] = list[n - 1 - idx - l];
    return L;
}

struct OctPlan { int L; int side[sttm::kOctMaxLevels]; size_t off_feat[sttm::kOctMaxLevels], off_stop[sttm::kOctMaxLevels], off_mark, off_lvl, off_rows, off_scan, scan_bytes, total; };

int octree_plan(int B, int side, int C, int dtype, int root_level, OctPlan* p) {
    const int L = octree_levels(side, root_level, p->side);
    if (L < 0) return L;
    p->L = L;
    size_t o = 0;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t eb = elem_bytes(dtype);
    for (int l = 0; l < L - 1; ++l) {
        const size_t cells = (size_t)B * p->side[l] * p->side[l] * p->side[l];
        p->off_feat[l] = o; o = al(o + cells * C * eb);
        p->off_stop[l] = o; o = al(o + cells);
    }
    const size_t leaves = (size_t)B * side * side * side;
    p->off_mark = o; o = al(o + leaves * 4);
    p->off_lvl = o; o = al(o + leaves);
    p->off_rows = o; o = al(o + leaves * 4);
    p->scan_bytes = sttm::octree_scan_bytes((int64_t)leaves);
    p->off_scan = o; o = al(o + p->scan_bytes);
    p->total = o;
    return L;
}

// Group-mean workgroups per frame: enough 4-wave workgroups (T * split) to cover the chip several times over.
int gm_split_for(int T, int HW, int nv = 1) {
    const int env = config().gm_split;
    // launch sets of several videos (the batch entry point: other streams' kernels run beside this one): a quarter of the workgroups
    // per frame -- sustained driver-style runs on one box, 3 streams x 8 videos: 18.2 k videos/s with 32 per frame and 1024-thread label
    // columns, 19.6 k with 16 / 512, 20.4 k with 8 / 512, 20.5 k with 8 / 256 (profiles/r05v_batch_switches.txt)
    const int want = env > 0 ? env : (nv > 1 ? (1024 + T - 1) / T : (4096 + T - 1) / T);
    int s = 1;
    while (s < want && s < 64) s <<= 1;          // a power of two: the kernel masks instead of dividing
    // frames of 512 and more tokens: every workgroup of a frame ranks ALL its survivors before it takes its share, so fewer, longer
    // shares are cheaper there (same-box sweep at T = 128: 20 x 36 tokens K5 36.3 -> 34.2 us with 16 instead of 32 per frame; 14 x 14 and
    // 13 x 24 lose 1.1-1.3 us with 16, so they keep 32)
    if (env <= 0 && HW >= 512 && s > 16) s = 16;
    return s;
}

// Pack width of the row-streaming kernels (pairs, group mean): they only read / write whole [*, C] rows, so fp32 can use
// 8-wide packs (two 16-byte loads in flight per lane; measured 23.7 -> 20.0 us for the group mean) even though the spatial
// kernel keeps 4-wide packs for occupancy.  Per-head cosine keeps the spatial kernel's width (head lanes are defined on it).
int row_vec(int C, int dtype, int spatial_vec, bool dense_x, bool aligned32, int head_dim) {
    if (dtype != STTM_F32 && spatial_vec == 16) return 8;          // the row kernels have no 32-byte 16-bit packs
    if (dtype == STTM_F32 && spatial_vec == 8) return (C % 8 == 0 && !(dense_x && !aligned32)) ? 8 : 4;
    if (dtype != STTM_F32 || head_dim != 0 || spatial_vec != 4) return spatial_vec;
    if (C % 8 || C < 512) return spatial_vec;
    if (dense_x && !aligned32) return spatial_vec;
    return 8;
}

void mark(void* const* events, int i, hipStream_t s) {
    if (events && events[i]) hipEventRecord(reinterpret_cast<hipEvent_t>(events[i]), s);
}

// a caller's node list for the stand-alone temporal stage (sttm_temporal_merge)
struct NodeList { const void* feat; const int32_t* tlbr; int n; };
// the unpooled token map of sttm_quadtree_merge_pooled: x is [T, src_h * src_w, C]; H x W of the call is the pooled grid
struct PoolSrc { int src_h, src_w, mode, stride; };

// The merge of up to kBatchMax same-shaped videos in one set of launches.
int merge_group(int nv, const void* const* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                int T, int C, int H, int W, int dtype,
                float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                void* workspace, size_t workspace_stride,
                void* const* feat_out, int32_t* const* npatch_out, int32_t* const* tlbr_out, int32_t* counts,
                int32_t* counts_host, int seq, void* const* events, hipStream_t stream,
                uint64_t* early_host = nullptr, int* n_early = nullptr, int flags = 0, int32_t* idx_out = nullptr,
                const NodeList* nodes = nullptr, const PoolSrc* pool = nullptr, int concurrent_sets = 1) {
    if (n_early) *n_early = 0;
    if (!x || !workspace || !feat_out || !npatch_out || !tlbr_out || !counts) return fail(STTM_ERR_ARG, "null pointer argument");
    if (nodes && nv != 1) return fail(STTM_ERR_UNSUPPORTED, "the stand-alone temporal stage takes one video");
    if (T < 1 || C < 1) return fail(STTM_ERR_ARG, "T and C must be positive");
    if (dtype < 0 || dtype > 2) return fail(STTM_ERR_ARG, "unknown dtype code %d", dtype);
    if (stride_c != 1) return fail(STTM_ERR_ARG, "channel stride must be 1 (channels-last view); got %lld", (long long)stride_c);
    if (H > 255 || W > 255) return fail(STTM_ERR_UNSUPPORTED, "token grids larger than 255 per side are not supported");
    if ((int64_t)T * H * W >= (1ll << 31) / 16) return fail(STTM_ERR_UNSUPPORTED, "too many tokens");
    if (head_dim < 0 || (head_dim > 0 && C % head_dim)) return fail(STTM_ERR_ARG, "head_dim %d does not divide C = %d", head_dim, C);
    Plan p;
    const int D = make_plan(T, H, W, C, dtype, root_level, &p);
    if (D < 0) return D;
    if (workspace_stride < p.bytes) return fail(STTM_ERR_ARG, "workspace too small: %zu < %zu", workspace_stride, p.bytes);
    if (reinterpret_cast<uintptr_t>(workspace) % 256 || (nv > 1 && workspace_stride % 256))
        return fail(STTM_ERR_ARG, "workspace (and the per-video stride) must be 256-byte aligned");
    if (weighted_avg && !nodes) {           // (the sum-pool pyramid of the spatial stage; the temporal stage weights by box areas only)
        for (int l = 1; l < D; ++l)
            if ((p.dims.h[l] & 1) != (p.dims.w[l] & 1))
                return fail(STTM_ERR_PARITY, "weighted_avg needs equal parities at every pooled level; level %dx%d is mixed",
                            p.dims.h[l], p.dims.w[l]);
    }
    if (p.max_area > 32767) return fail(STTM_ERR_UNSUPPORTED, "root cells of %d leaves (limit 32767)", p.max_area);
    // multiply-high divisions of the label stage: j / ecap for j < (T-1)*ecap needs (T-1) * ecap^2 <= 2^32
    if ((double)(T > 1 ? T - 1 : 1) * p.ecap * (double)p.ecap > 4294967296.0)
        return fail(STTM_ERR_UNSUPPORTED, "T * (root-cell area)^2 too large for the label stage (T=%d, area=%d)", T, p.max_area);
    sttm::BatchPtrs bp;
    memset(&bp, 0, sizeof(bp));
    bp.ws_stride = workspace_stride;
    uintptr_t all_bits = 0;
    for (int v = 0; v < nv; ++v) {
        if (!x[v] || !feat_out[v] || !npatch_out[v] || !tlbr_out[v]) return fail(STTM_ERR_ARG, "null per-video pointer (video %d)", v);
        bp.x[v] = x[v]; bp.feat[v] = feat_out[v]; bp.npatch[v] = npatch_out[v]; bp.tlbr[v] = tlbr_out[v];
        all_bits |= reinterpret_cast<uintptr_t>(x[v]);
    }
    const void* x_align = reinterpret_cast<const void*>(all_bits);      // the least aligned of the inputs decides the pack width
    int nt = 0;
    if (int rc = check_frame_addressing(pool ? pool->src_h : H, pool ? pool->src_w : W, C, (int)elem_bytes(dtype), stride_h, stride_w, stride_t)) return rc;
    const int vec = pick_vec(C, dtype, x_align, stride_t, stride_h, stride_w, &nt, head_dim == 0 && !pool);
    if (pool) {
        // the fused pooled-leaf load exists for the shape the production presets use: 3-level trees, 16-byte packs, whole-vector cosine
        if (p.dims.n_level != 3 || head_dim != 0 || vec * (int)elem_bytes(dtype) != 16 || H > sttm::kPoolMaxSide || W > sttm::kPoolMaxSide)
            return fail(STTM_ERR_UNSUPPORTED, "pooled input: the fused form needs a 3-level tree (got %d levels), the whole-vector cosine and 16-byte "
                        "aligned rows with C %% %d == 0, a pooled grid of at most %d x %d; pool with sttm_pool2d and call sttm_quadtree_merge instead",
                        p.dims.n_level, 16 / (int)elem_bytes(dtype), sttm::kPoolMaxSide, sttm::kPoolMaxSide);
    }
    if (!vec) return fail(STTM_ERR_UNSUPPORTED, "C=%d with this alignment does not fit one workgroup (need C/vec <= 1024 lanes)", C);
    int n_head = 0, head_lanes = 0;
    if (head_dim > 0) {
        // one head = head_dim / vec adjacent lanes: must be a power of two that fits a wave
        if (head_dim % vec) return fail(STTM_ERR_UNSUPPORTED, "head_dim %d is not a multiple of the %d-wide channel pack", head_dim, vec);
        head_lanes = head_dim / vec;
        if (head_lanes > 64 || (head_lanes & (head_lanes - 1)))
            return fail(STTM_ERR_UNSUPPORTED, "head_dim / pack width = %d lanes: need a power of two <= 64", head_lanes);
        n_head = C / head_dim;
    }
    const Config& cfg = config();
    Buffers b;
    carve_all(p, T, H, W, C, dtype, reinterpret_cast<char*>(workspace), &b);
    sttm::SpatialArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.x = x[0]; sa.sT = stride_t; sa.sH = stride_h; sa.sW = stride_w;
    sa.T = T; sa.H = H; sa.W = W; sa.C = C;
    sa.dims = p.dims;
    sa.threshold = threshold;
    sa.thr_lo_sq = thr_lo_sq_of(threshold);
    sa.sum_mode = weighted_avg ? 1 : 0;
    sa.n_head = n_head; sa.head_lanes = head_lanes;
    // dense [T*H*W, C] input: the rows of 1x1 nodes are read from x by the later kernels instead of being copied to S
    const bool dense = !nodes && !pool && stride_w == C && stride_h == (int64_t)W * C && stride_t == (int64_t)H * W * C;
    if (pool) {
        sa.src_h = pool->src_h; sa.src_w = pool->src_w; sa.pool_mode = pool->mode;
        sa.pool_sh = (float)pool->src_h / (float)H; sa.pool_sw = (float)pool->src_w / (float)W;
    }
    sa.leaves_in_x = dense ? 1 : 0;
    sa.k1_var = cfg.k1_var;
    sa.S = b.S; sa.meta = b.meta; sa.inrm = b.inrm; sa.rc_list = b.rc_list;
    sa.rc_stride = p.rc_stride;
    sa.lab_row = b.lab_row; sa.gcnt = b.gcnt; sa.cgeo = b.cgeo;
    sa.counts = counts;
    sa.frame_cnt = b.frame_cnt;
    sa.bar = b.bar;
    sa.col_arrive = b.col_arrive;

    sttm::TemporalArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.T = T; ta.H = H; ta.W = W; ta.C = C; ta.R = p.R;
    ta.dims = p.dims;
    ta.dtype = dtype; ta.vec = row_vec(C, dtype, vec, dense, all_bits % 32 == 0, head_dim);
    // The column-walk spatial stage (spatial_col.inc, opt-in: col_walk) forms the pair stage's dot products itself; it covers the production
    // shape -- 3-level trees, whole-vector cosine, the fast pair filter, 16-byte packs of rows of at most 512 lanes -- and is meant for
    // launch sets of several videos (one video has too few columns to fill the device).  So that both forms give the same bits, the pair
    // kernel of such a shape sums in the column walk's order: the spatial kernel's 16-byte packs.
    sttm::ColWalkArgs cw;
    memset(&cw, 0, sizeof(cw));
    const int vec16 = 16 / (int)elem_bytes(dtype);
    bool col_shape = !nodes && !pool && p.dims.n_level == 3 && head_dim == 0 && !slow_ver && temporal_thresh > 0.f && T > 1 &&
                     vec * (int)elem_bytes(dtype) >= 16 && C % vec16 == 0 && (C / vec16 + 63) / 64 * 64 <= 512;
    int col_nt = 0;
    if (col_shape) {
        col_nt = (C / vec16 + 63) / 64 * 64;
        const size_t budget = (col_nt <= 256 ? 53 : 80) * 1024;                 // three (two) workgroups per CU
        cw.pb = cfg.col_pb > 0 ? cfg.col_pb : (col_nt <= 256 ? 10 : 6);
        if (cw.pb > 16) cw.pb = 16;
        const size_t fixed = sttm::col_walk_lds_bytes(col_nt, 0, cw.pb);
        int cap = fixed < budget ? (int)((budget - fixed) / ((size_t)col_nt * 16)) : 0;
        if (cap > 16) cap = 16;
        if (cfg.col_cap > 0 && cfg.col_cap < cap) cap = cfg.col_cap;
        cw.cap = cap;
        cw.frames = cfg.col_frames > 0 ? cfg.col_frames : 8;
        cw.abl = cfg.col_abl;
        if (cap < 4 && cfg.col_cap <= 0) col_shape = false;
    }
    const bool col_walk = col_shape && (cfg.col_walk >= 2 || (cfg.col_walk == 1 && nv > 1));
    ta.pair_vec = cfg.pair_vec > 0 ? cfg.pair_vec : ((col_shape && cfg.pair_vec == 0) ? vec16 : ta.vec);
    sttm::pairs_shape(T, p.R, cfg.fold_labels && !slow_ver, cfg.pairs_seg, cfg.pairs_nt, &ta.pairs_seg, &ta.pairs_nt);
    ta.pairs_var = cfg.pairs_var;
    // threads per label column: 1024 for one video (the stage is the call's critical path), 256 in launch sets of several videos (the
    // stage then runs beside other streams' bandwidth-bound kernels and should leave them the CUs)
    ta.label_nt = (cfg.label_nt == 256 || cfg.label_nt == 512 || cfg.label_nt == 1024) ? cfg.label_nt : (nv > 1 ? 256 : 1024);
    ta.temporal_thresh = temporal_thresh;
    // the merge runs its temporal stage only for a positive threshold (quadtree_builder.py:217); cross_frame_node_merging_fast / _slow
    // themselves filter with whatever threshold they are given (quadtree_temporal_merger.py:70-71: sim >= thresh)
    ta.temporal_on = (temporal_thresh > 0.f || nodes) ? 1 : 0;
    ta.weighted_avg = weighted_avg ? 1 : 0;
    // slow_ver has no per-head variant upstream (cross_frame_node_merging_slow ignores head_dim)
    ta.n_head = slow_ver ? 0 : n_head; ta.head_lanes = slow_ver ? 0 : head_lanes;
    ta.inline_norms = (slow_ver && n_head > 0) ? 1 : 0;
    ta.slow_ver = slow_ver ? 1 : 0;
    ta.max_slots = p.max_slots;
    ta.force_gmem = cfg.force_gmem_labels ? 1 : 0;
    ta.no_fuse = (cfg.no_fuse || (flags & STTM_FLAG_NO_FUSE)) ? 1 : 0;
    ta.no_dense = cfg.no_dense;          // 0: dense form (round 4), 1: compact ids, 2: round 3's dense form (A/B)
    ta.want_fold = cfg.fold_labels ? 1 : 0;
    ta.fold_kb = cfg.fold_kb > 0 ? cfg.fold_kb : 64;
    ta.S = b.S; ta.xrows = dense ? x[0] : nullptr; ta.meta = b.meta; ta.inrm = b.inrm; ta.rc_list = b.rc_list; ta.rc_stride = p.rc_stride;
    ta.edges = b.edges; ta.edge_sim = slow_ver ? b.edge_sim : nullptr; ta.ecap = p.ecap; ta.edge_cnt = b.edge_cnt; ta.cand_cnt = b.cand_cnt;
    ta.ecap_magic = 0xffffffffu / (unsigned)p.ecap + 1u;
    ta.col_mask = b.col_mask; ta.col_arrive = b.col_arrive; ta.frame_cnt = b.frame_cnt; ta.bar = b.bar;
    ta.colscratch = b.colscratch;
    ta.gm_split = gm_split_for(T, H * W, nv); ta.lab_row = b.lab_row; ta.gcnt = b.gcnt; ta.cgeo = b.cgeo;
    ta.counts = counts;
    ta.counts_host = counts_host; ta.seq = seq;
    // every column reports its survivors to the host itself (single video, a slot per column)
    // (a batch: video v reports into early_host[v * STTM_EARLY_SLOTS ...], sequence number seq + v)
    if (early_host && n_early && p.R <= STTM_EARLY_SLOTS) {
        ta.early_host = reinterpret_cast<unsigned long long*>(early_host);
        *n_early = p.R;
    }
    ta.feat_out = feat_out[0]; ta.npatch_out = npatch_out[0]; ta.tlbr_out = tlbr_out[0];
    ta.idx_out = nv == 1 ? idx_out : nullptr;
#ifdef STTM_DEV
    sa.dev = g_dev; ta.dev = g_dev;
#endif

    const bool pairs = ta.temporal_on && T > 1;
    int fold_cap = 0;
    ta.fold_labels = (pairs && !col_walk && sttm::labels_can_fold(ta, nv, &fold_cap)) ? 1 : 0;
    ta.fold_cap = fold_cap;

    hipError_t e;
    mark(events, 0, stream);
    // trees of 4 and more levels (k1_split = -1: never; 5: from 5 levels): one workgroup per 3-level block + a pass over the upper levels
    const int split_from = cfg.k1_split < 0 ? 99 : (cfg.k1_split == 5 ? 5 : 4);
    void* tops = (n_head == 0 && p.dims.n_level >= split_from) ? b.tops : nullptr;
    // the one-workgroup form of a 6-level tree (per-head cosine, k1_split = -1) keeps 2735 partial statistics per wave in LDS
    // (10.7 KB): at most 12 waves fit next to the other tables
    if (!tops && p.dims.n_level >= 6 && nt > 768)
        return fail(STTM_ERR_UNSUPPORTED, "a %d-level tree with %d lanes per token row does not fit the LDS in the one-workgroup form "
                    "(per-head cosine or k1_split = -1; 6-level trees: <= 768 lanes, i.e. fp32 C <= 3072, 16-bit C <= 6144)", p.dims.n_level, nt);
    if (nodes) {
        if ((e = sttm::launch_ingest_nodes(ta, nodes->feat, nodes->tlbr, nodes->n, b.S, b.meta, b.inrm, b.rc_list, b.cgeo, stream)) != hipSuccess)
            return fail(STTM_ERR_LAUNCH, "node ingest: %s", hipGetErrorString(e));
    } else if (pool) {
        if ((e = sttm::launch_spatial_pooled(sa, bp, nv, dtype, nt, stream)) != hipSuccess)
            return fail(STTM_ERR_LAUNCH, "spatial kernel (pooled input): %s", hipGetErrorString(e));
    } else if (col_walk) {
        cw.edges = ta.edges; cw.edge_cnt = ta.edge_cnt; cw.cand_cnt = ta.cand_cnt; cw.ecap = ta.ecap; cw.temporal_thresh = temporal_thresh;
        if ((e = sttm::launch_spatial_col(sa, bp, cw, nv, dtype, col_nt, stream)) != hipSuccess)
            return fail(STTM_ERR_LAUNCH, "spatial kernel (column walk): %s", hipGetErrorString(e));
    } else if ((e = sttm::launch_spatial(sa, bp, nv, dtype, vec, nt, stream, tops)) != hipSuccess)
        return fail(STTM_ERR_LAUNCH, "spatial kernel: %s", hipGetErrorString(e));
    mark(events, 1, stream);
#ifdef STTM_DEV
    if (g_dev.k1_mode == 1 || g_dev.k1_mode == 2) {      // ablation run: only the spatial kernel is launched, outputs are not valid
        for (int i = 2; i < STTM_EVENT_SLOTS; ++i) mark(events, i, stream);
        return STTM_OK;
    }
#endif
    if (pairs && !col_walk) {
        if ((e = sttm::launch_pairs(ta, bp, nv, stream)) != hipSuccess)
            return fail(STTM_ERR_LAUNCH, "pairs kernel: %s", hipGetErrorString(e));
    }
    if (pairs && slow_ver) {
        if ((e = sttm::launch_slow_filter(ta, bp, nv, stream)) != hipSuccess)
            return fail(e == hipErrorInvalidValue ? STTM_ERR_UNSUPPORTED : STTM_ERR_LAUNCH, "slow_ver filter kernel: %s", hipGetErrorString(e));
    }
    mark(events, 2, stream);
    if (!ta.fold_labels) {
        // (the stage-skewed batch form runs `concurrent_sets` launch sets at once, and every one of them can sit in the fused label
        // kernel's grid barrier at the same time: the residency budget is shared between them)
        if (sttm::labels_can_fuse(ta, nv, concurrent_sets)) {
            if ((e = sttm::launch_labels_fused(ta, bp, nv, stream)) != hipSuccess)
                return fail(STTM_ERR_LAUNCH, "fused label kernel: %s", hipGetErrorString(e));
        } else if ((e = sttm::launch_col_labels(ta, bp, nv, true, stream)) != hipSuccess ||
                   (e = sttm::launch_col_labels(ta, bp, nv, false, stream)) != hipSuccess)
            return fail(STTM_ERR_LAUNCH, "label kernels: %s", hipGetErrorString(e));
    }
    mark(events, 3, stream);
    if ((e = sttm::launch_group_mean(ta, bp, nv, stream)) != hipSuccess)
        return fail(STTM_ERR_LAUNCH, "group-mean kernel: %s", hipGetErrorString(e));
    mark(events, 4, stream);
    return STTM_OK;
}

// Internal streams of the stage-skewed batch entry point: per host thread and device (two host threads may be inside
// sttm_quadtree_merge_batch at once, each on its own caller stream; sharing the fork / join events between them would race), created
// on first use and kept for the life of the thread (HIP may already be gone when thread-locals are destroyed at exit: never freed).
constexpr int kSideMax = 8;
struct SidePool { int dev; int n; hipStream_t s[kSideMax]; hipEvent_t fork; hipEvent_t join[kSideMax]; };
// the calling thread's pool for the current device with at least `want` streams (grown on demand); nullptr if HIP refuses
thread_local SidePool g_side_pools[4] = {};
thread_local int g_side_pool_count = 0;
SidePool* side_pool(int want) {
    SidePool* const pools = g_side_pools;
    int& n_pools = g_side_pool_count;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (want > kSideMax) want = kSideMax;
    SidePool* p = nullptr;
    for (int i = 0; i < n_pools; ++i)
        if (pools[i].dev == dev) p = &pools[i];
    if (!p) {
        if (n_pools >= 4) return nullptr;
        p = &pools[n_pools];
        *p = SidePool{};
        p->dev = dev;
        if (hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        ++n_pools;
    }
    while (p->n < want) {
        if (hipStreamCreateWithFlags(&p->s[p->n], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&p->join[p->n], hipEventDisableTiming) != hipSuccess) return nullptr;
        ++p->n;
    }
    return p;
}

}  // namespace

namespace sttm {
int tome_split_mode() { return config().tome_split; }
int tome_flat_mode() { return config().tome_flat; }
}  // namespace sttm

extern "C" {

#ifndef STTM_BUILD_TAG
#define STTM_BUILD_TAG "untagged"
#endif
int sttm_abi_version(void) { return STTM_ABI_VERSION; }
const char* sttm_build_tag(void) { return STTM_BUILD_TAG; }
const char* sttm_last_error(void) { return g_err; }

int sttm_release_streams(void) {
    // the calling thread's internal streams / events of sttm_quadtree_merge_batch (all devices): synchronised, destroyed, forgotten
    int n = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int i = 0; i < g_side_pool_count; ++i) {
        SidePool& p = g_side_pools[i];
        (void)hipSetDevice(p.dev);
        for (int k = 0; k < p.n; ++k) {
            (void)hipStreamSynchronize(p.s[k]);
            (void)hipStreamDestroy(p.s[k]);
            (void)hipEventDestroy(p.join[k]);
            ++n;
        }
        (void)hipEventDestroy(p.fork);
        p = SidePool{};
    }
    g_side_pool_count = 0;
    (void)hipSetDevice(cur);
    return n;
}

int sttm_configure(const char* key, int value) {
    if (!key) return fail(STTM_ERR_ARG, "null key");
    Config& c = config();
    struct { const char* name; int* slot; } keys[] = {
        {"pairs_seg", &c.pairs_seg}, {"pairs_nt", &c.pairs_nt}, {"pairs_var", &c.pairs_var}, {"k1_var", &c.k1_var}, {"k1_split", &c.k1_split}, {"no_dense", &c.no_dense}, {"gm_split", &c.gm_split}, {"label_nt", &c.label_nt},
        {"vec16", &c.vec16}, {"vec32", &c.vec32}, {"fold_kb", &c.fold_kb}, {"fold_labels", &c.fold_labels}, {"no_fuse", &c.no_fuse}, {"tome_split", &c.tome_split}, {"tome_flat", &c.tome_flat},
        {"force_gmem_labels", &c.force_gmem_labels}, {"batch_streams", &c.batch_streams}, {"batch_sub", &c.batch_sub},
        {"col_walk", &c.col_walk}, {"col_frames", &c.col_frames}, {"col_cap", &c.col_cap}, {"col_pb", &c.col_pb},
#ifdef STTM_DEV
        {"col_abl", &c.col_abl},
#endif
        {"pair_vec", &c.pair_vec},
    };
    for (auto& k : keys)
        if (!strcmp(key, k.name)) { *k.slot = value; return STTM_OK; }
    return fail(STTM_ERR_ARG, "unknown configuration key '%s'", key);
}

int sttm_quadtree_num_levels(int H, int W, int root_level) {
    sttm::LevelDims d;
    return build_dims(H, W, root_level, &d);
}

size_t sttm_quadtree_workspace_bytes(int T, int H, int W, int C, int dtype, int root_level) {
    if (T < 1 || C < 1 || dtype < 0 || dtype > 2) { fail(STTM_ERR_ARG, "bad T/C/dtype"); return 0; }
    Plan p;
    if (make_plan(T, H, W, C, dtype, root_level, &p) < 0) return 0;
    return p.bytes;
}

int sttm_quadtree_merge(const void* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                        int T, int C, int H, int W, int dtype,
                        float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                        void* workspace, size_t workspace_bytes,
                        void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                        void* stream_) {
    return sttm_quadtree_merge_async(x, stride_t, stride_c, stride_h, stride_w, T, C, H, W, dtype, threshold, temporal_thresh,
                                     root_level, weighted_avg, head_dim, slow_ver, workspace, workspace_bytes, feat_out, npatch_out,
                                     tlbr_out, counts, nullptr, 0, nullptr, stream_);
}

int sttm_quadtree_spatial(const void* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                          int T, int C, int H, int W, int dtype, float threshold, int root_level, int weighted_avg, int head_dim,
                          void* workspace, size_t workspace_bytes, void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                          void* stream_) {
    // the spatial stage alone = the merge with the temporal stage switched off (quadtree_builder.py:217: `if temporal_thresh > 0`)
    return sttm_quadtree_merge(x, stride_t, stride_c, stride_h, stride_w, T, C, H, W, dtype, threshold, -1.0f, root_level, weighted_avg,
                               head_dim, 0, workspace, workspace_bytes, feat_out, npatch_out, tlbr_out, counts, stream_);
}

int sttm_temporal_merge(const void* node_feat, const int32_t* node_tlbr, int n_nodes, int T, int C, int H, int W, int dtype,
                        float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                        void* workspace, size_t workspace_bytes, void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                        void* stream_) {
    // cross_frame_node_merging_fast / _slow on a caller's node list (quadtree_temporal_merger.py:271-299): the nodes are brought into
    // the layout of the spatial stage's outputs, then the pair, label and group-mean kernels of the full merge run unchanged
    if (!node_feat || !node_tlbr) return fail(STTM_ERR_ARG, "null node list");
    if (n_nodes < 0 || (int64_t)n_nodes > (int64_t)T * H * W) return fail(STTM_ERR_ARG, "n_nodes = %d outside [0, T*H*W]", n_nodes);
    const NodeList nodes{node_feat, node_tlbr, n_nodes};
    return merge_group(1, &node_feat, (int64_t)H * W * C, 1, (int64_t)W * C, C, T, C, H, W, dtype, 2.0f, temporal_thresh, root_level,
                       weighted_avg, head_dim, slow_ver, workspace, workspace_bytes, &feat_out, &npatch_out, &tlbr_out, counts,
                       nullptr, 0, nullptr, reinterpret_cast<hipStream_t>(stream_), nullptr, nullptr, 0, nullptr, &nodes);
}

int sttm_quadtree_merge_async(const void* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                              int T, int C, int H, int W, int dtype,
                              float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                              void* workspace, size_t workspace_bytes,
                              void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                              int32_t* counts_host, int seq, void* const* events, void* stream_) {
    return merge_group(1, &x, stride_t, stride_c, stride_h, stride_w, T, C, H, W, dtype, threshold, temporal_thresh, root_level,
                       weighted_avg, head_dim, slow_ver, workspace, workspace_bytes, &feat_out, &npatch_out, &tlbr_out, counts,
                       counts_host, seq, events, reinterpret_cast<hipStream_t>(stream_));
}

int sttm_quadtree_merge_packed(sttm_merge_args* g) {
    if (!g) return fail(STTM_ERR_ARG, "null argument block");
    const void* x = g->x;
    void* feat = g->feat_out;
    int32_t* np = g->npatch_out;
    int32_t* tl = g->tlbr_out;
    int n_early = 0;
    const int rc = merge_group(1, &x, g->stride_t, g->stride_c, g->stride_h, g->stride_w, g->T, g->C, g->H, g->W, g->dtype, g->threshold,
                               g->temporal_thresh, g->root_level, g->weighted_avg, g->head_dim, g->slow_ver, g->workspace, g->workspace_bytes,
                               &feat, &np, &tl, g->counts, g->counts_host, g->seq, g->events, reinterpret_cast<hipStream_t>(g->stream),
                               g->early_host, &n_early, g->flags, g->idx_out);
    g->n_early = n_early;
    return rc;
}

int sttm_quadtree_merge_pooled(const void* x_tokens, int T, int src_h, int src_w, int C, int dtype, int pool_mode, int pool_stride,
                               float threshold, float temporal_thresh, int root_level, int weighted_avg, int slow_ver,
                               void* workspace, size_t workspace_bytes,
                               void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                               int32_t* counts_host, int seq, void* stream_, int flags) {
    if (src_h < 1 || src_w < 1 || pool_stride < 1) return fail(STTM_ERR_ARG, "bad source grid / stride");
    if (pool_mode < STTM_POOL_AVERAGE || pool_mode > STTM_POOL_BILINEAR) return fail(STTM_ERR_ARG, "Unexpected mm_spatial_pool_mode: %d", pool_mode);
    if (pool_stride == 1) return fail(STTM_ERR_ARG, "stride 1 is the identity (get_2dPool returns its input): call sttm_quadtree_merge");
    if (pool_mode != STTM_POOL_BILINEAR && pool_stride != 2)
        return fail(STTM_ERR_UNSUPPORTED, "the fused form of the average / max pool is the 2 x 2 window (stride 2); pool with sttm_pool2d instead");
    const int H = sttm_pool2d_out_side(src_h, pool_stride, pool_mode), W = sttm_pool2d_out_side(src_w, pool_stride, pool_mode);
    if (H < 1 || W < 1) return fail(STTM_ERR_ARG, "pooling window larger than the grid");
    const PoolSrc pool{src_h, src_w, pool_mode, pool_stride};
    return merge_group(1, &x_tokens, (int64_t)src_h * src_w * C, 1, (int64_t)src_w * C, C, T, C, H, W, dtype, threshold, temporal_thresh,
                       root_level, weighted_avg, 0, slow_ver, workspace, workspace_bytes, &feat_out, &npatch_out, &tlbr_out, counts,
                       counts_host, seq, nullptr, reinterpret_cast<hipStream_t>(stream_), nullptr, nullptr, flags, nullptr, nullptr, &pool);
}

int sttm_quadtree_merge_batch(int n_videos, const void* const* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                              int T, int C, int H, int W, int dtype,
                              float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                              void* workspace, size_t workspace_stride,
                              void* const* feat_out, int32_t* const* npatch_out, int32_t* const* tlbr_out, int32_t* counts,
                              int32_t* counts_host, int seq, void* const* events, void* stream_, int flags,
                              uint64_t* early_host, int* n_early_out) {
    if (n_early_out) *n_early_out = 0;
    if (n_videos < 1) return fail(STTM_ERR_ARG, "n_videos must be >= 1");
    if (!x || !feat_out || !npatch_out || !tlbr_out || !counts || !workspace) return fail(STTM_ERR_ARG, "null pointer argument");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const Config& cfg = config();
    // Stage-skewed form (round 5): the videos of the call are dealt out to a few internal streams, each running whole per-video
    // chains (spatial -> pairs -> labels -> group mean) behind a fork event on the caller's stream; the caller's stream joins them
    // at the end.  Consecutive videos are then in DIFFERENT stages at any moment -- the latency-bound label stage (16 workgroups)
    // and the ramps / tails of one video run under the bandwidth-bound kernels of the others -- where the lockstep form below
    // marches all videos through one kernel at a time.  Same kernels, same per-video arguments: bit-identical outputs.
    SidePool* pool = (cfg.batch_streams >= 2 && n_videos >= 2) ? side_pool(cfg.batch_streams) : nullptr;
    if (pool) {
        const int want = cfg.batch_streams < kSideMax ? cfg.batch_streams : kSideMax;
        const int S = want < n_videos ? want : n_videos;
        const int sub = cfg.batch_sub < 1 ? 1 : (cfg.batch_sub > STTM_BATCH_MAX ? STTM_BATCH_MAX : cfg.batch_sub);
        mark(events, 0, stream);
        hipError_t e = hipEventRecord(pool->fork, stream);
        for (int i = 0; i < S && e == hipSuccess; ++i) e = hipStreamWaitEvent(pool->s[i], pool->fork, 0);
        if (e != hipSuccess) return fail(STTM_ERR_LAUNCH, "batch fork: %s", hipGetErrorString(e));
        int rc = STTM_OK, lane = 0;
        for (int v0 = 0; v0 < n_videos && rc == STTM_OK; v0 += sub, lane = (lane + 1) % S) {
            const int nv = n_videos - v0 < sub ? n_videos - v0 : sub;
            rc = merge_group(nv, x + v0, stride_t, stride_c, stride_h, stride_w, T, C, H, W, dtype, threshold, temporal_thresh,
                             root_level, weighted_avg, head_dim, slow_ver,
                             reinterpret_cast<char*>(workspace) + (size_t)v0 * workspace_stride, workspace_stride,
                             feat_out + v0, npatch_out + v0, tlbr_out + v0, counts + (size_t)v0 * STTM_CNT_SLOTS,
                             counts_host ? counts_host + (size_t)v0 * STTM_CNT_SLOTS : nullptr, seq + v0,
                             nullptr, pool->s[lane], early_host ? early_host + (size_t)v0 * STTM_EARLY_SLOTS : nullptr,
                             early_host ? n_early_out : nullptr, flags, nullptr, nullptr, nullptr, S);
        }
        // join even after a failed launch: whatever was enqueued on the internal streams stays ordered before the caller's next work
        for (int i = 0; i < S; ++i) {
            hipError_t j = hipEventRecord(pool->join[i], pool->s[i]);
            if (j == hipSuccess) j = hipStreamWaitEvent(stream, pool->join[i], 0);
            if (j != hipSuccess && rc == STTM_OK) rc = fail(STTM_ERR_LAUNCH, "batch join: %s", hipGetErrorString(j));
        }
        for (int i = 1; i < STTM_EVENT_SLOTS; ++i) mark(events, i, stream);
        return rc;
    }
    for (int v0 = 0; v0 < n_videos; v0 += STTM_BATCH_MAX) {
        const int nv = n_videos - v0 < STTM_BATCH_MAX ? n_videos - v0 : STTM_BATCH_MAX;
        // the caller's events bracket the whole call: the start comes from the first group, the rest from the last
        void* ev[STTM_EVENT_SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (events) {
            if (v0 == 0) ev[0] = events[0];
            if (v0 + nv >= n_videos)
                for (int i = 1; i < STTM_EVENT_SLOTS; ++i) ev[i] = events[i];
        }
        const int rc = merge_group(nv, x + v0, stride_t, stride_c, stride_h, stride_w, T, C, H, W, dtype, threshold, temporal_thresh,
                                   root_level, weighted_avg, head_dim, slow_ver,
                                   reinterpret_cast<char*>(workspace) + (size_t)v0 * workspace_stride, workspace_stride,
                                   feat_out + v0, npatch_out + v0, tlbr_out + v0, counts + (size_t)v0 * STTM_CNT_SLOTS,
                                   counts_host ? counts_host + (size_t)v0 * STTM_CNT_SLOTS : nullptr, seq + v0,
                                   events ? ev : nullptr, stream, early_host ? early_host + (size_t)v0 * STTM_EARLY_SLOTS : nullptr,
                                   early_host ? n_early_out : nullptr, flags);
        if (rc != STTM_OK) return rc;
    }
    return STTM_OK;
}

int sttm_quadtree_apply(const void* v, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                        int T, int Cv, int H, int W, int dtype_v, int sum_mode,
                        int C_feat, int dtype_feat, int root_level, void* workspace, size_t workspace_bytes,
                        const int32_t* counts, void* out, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!v || !workspace || !counts || !out) return fail(STTM_ERR_ARG, "null pointer argument");
    if (stride_c != 1) return fail(STTM_ERR_ARG, "channel stride must be 1 (channels-last view)");
    if (dtype_v < 0 || dtype_v > 2 || Cv < 1) return fail(STTM_ERR_ARG, "bad dtype / channel count");
    Plan p;
    const int D = make_plan(T, H, W, C_feat, dtype_feat, root_level, &p);
    if (D < 0) return D;
    if (workspace_bytes < p.bytes) return fail(STTM_ERR_ARG, "workspace too small");
    if ((size_t)Cv * elem_bytes(dtype_v) > (size_t)C_feat * elem_bytes(dtype_feat))
        return fail(STTM_ERR_UNSUPPORTED, "the side tensor is wider than the feature rows the scratch was sized for");
    for (int l = 1; l < D; ++l)
        if ((p.dims.h[l] & 1) != (p.dims.w[l] & 1))
            return fail(STTM_ERR_PARITY, "position-embedding pooling needs equal parities at every pooled level; level %dx%d is mixed",
                        p.dims.h[l], p.dims.w[l]);
    int nt = 0;
    if (int rc = check_frame_addressing(H, W, Cv, (int)elem_bytes(dtype_v), stride_h, stride_w, stride_t)) return rc;
    const int vec = pick_vec(Cv, dtype_v, v, stride_t, stride_h, stride_w, &nt);
    if (!vec) return fail(STTM_ERR_UNSUPPORTED, "channel count / alignment of the side tensor is not supported");
    Buffers b;
    carve_all(p, T, H, W, C_feat, dtype_feat, reinterpret_cast<char*>(workspace), &b);
    const bool dense = stride_w == Cv && stride_h == (int64_t)W * Cv && stride_t == (int64_t)H * W * Cv;
    sttm::SpatialArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.x = v; sa.sT = stride_t; sa.sH = stride_h; sa.sW = stride_w;
    sa.T = T; sa.H = H; sa.W = W; sa.C = Cv;
    sa.dims = p.dims;
    sa.sum_mode = sum_mode ? 1 : 0;
    sa.leaves_in_x = dense ? 1 : 0;
    sa.S = b.S; sa.rc_list = b.rc_list; sa.rc_stride = p.rc_stride;
    hipError_t e;
    if ((e = sttm::launch_node_apply(sa, dtype_v, vec, nt, stream)) != hipSuccess)
        return fail(STTM_ERR_LAUNCH, "node-apply kernel: %s", hipGetErrorString(e));
    sttm::TemporalArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.T = T; ta.H = H; ta.W = W; ta.C = Cv; ta.R = p.R;
    ta.dims = p.dims;
    ta.dtype = dtype_v; ta.vec = vec;
    ta.weighted_avg = sum_mode ? 1 : 0;
    ta.S = b.S; ta.xrows = dense ? v : nullptr;
    ta.frame_cnt = b.frame_cnt; ta.lab_row = b.lab_row; ta.gcnt = b.gcnt; ta.cgeo = b.cgeo;
    ta.meta = b.meta; ta.gm_split = gm_split_for(T, H * W);
    ta.counts = const_cast<int32_t*>(counts);
    ta.feat_out = out;
    sttm::BatchPtrs bp;
    memset(&bp, 0, sizeof(bp));
    bp.x[0] = v; bp.feat[0] = out;
    if ((e = sttm::launch_group_mean(ta, bp, 1, stream)) != hipSuccess)
        return fail(STTM_ERR_LAUNCH, "group-mean kernel: %s", hipGetErrorString(e));
    return STTM_OK;
}

// The wait for N': a short busy spin (the counts usually arrive within tens of microseconds), then a yielding poll so that
// a long kernel queue ahead of this call does not burn a core.
int sttm_wait_counts_early(const int32_t* counts_host, const uint64_t* early_host, int n_early, int seq, int timeout_us, int32_t* out) {
    if (!counts_host || !out || (n_early > 0 && !early_host) || n_early > STTM_EARLY_SLOTS) return fail(STTM_ERR_ARG, "bad arguments");
    const volatile int32_t* flag = counts_host + STTM_CNT_SLOTS - 1;
    const volatile uint64_t* col = early_host;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int have = 0;                         // columns 0 .. have-1 have reported (each word is written once per seq)
    long long sum = 0;
    unsigned flags = 0;
    for (unsigned spins = 0;; ++spins) {
        while (have < n_early) {
            const uint64_t w = col[have];
            if ((uint32_t)(w >> 32) != (uint32_t)seq) break;
            sum += (long long)(w & 0x0fffffffull);
            flags |= (unsigned)(w & 0xc0000000ull);
            ++have;
        }
        if (n_early > 0 && have == n_early) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            out[0] = (int32_t)sum;
            out[1] = ((flags & 0x80000000u) ? STTM_OVF_BARRIER_TIMEOUT : 0) | ((flags & 0x40000000u) ? 1 : 0);
            return STTM_OK;
        }
        if (*flag == seq) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            out[0] = counts_host[STTM_CNT_OUT];
            out[1] = counts_host[STTM_CNT_OVERFLOW];
            return STTM_OK;
        }
        __builtin_ia32_pause();
        if ((spins & 255u) == 255u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const long long us = (t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000;
            if (us > timeout_us) return STTM_ERR_TIMEOUT;
            if (us > 300) sched_yield();
        }
    }
}

int sttm_wait_counts(const int32_t* counts_host, int seq, int timeout_us) {
    if (!counts_host) return fail(STTM_ERR_ARG, "null pointer");
    const volatile int32_t* flag = counts_host + STTM_CNT_SLOTS - 1;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; ++spins) {
        if (*flag == seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return STTM_OK; }
        __builtin_ia32_pause();
        if ((spins & 255u) == 255u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const long long us = (t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000;
            if (us > timeout_us) return STTM_ERR_TIMEOUT;
            if (us > 300) sched_yield();
        }
    }
}

#ifdef STTM_DEV
// development build only (not in the public header): measurement hooks of tools/*_ticks.py and tools/k1_ablate.py.
// ticks: device buffer of >= 48 long longs -- [0, 16) spatial workgroup k1_wg, [16, 32) pair workgroup k2_wg, [32, 48) label
// stage of column lbl_col -- or NULL to switch the stamps off.
int sttm_dev_k1_span(long long* span) { g_dev.k1_span = span; return STTM_OK; }
int sttm_dev_k5_mode(int mode) { g_dev.k5_mode = mode; return STTM_OK; }   // [2 * T * R] device buffer or NULL
int sttm_dev_hooks(int k1_mode, long long* ticks, int k1_wg, int k2_wg, int lbl_col) {
    g_dev.k1_mode = k1_mode;
    g_dev.k1_ticks = ticks; g_dev.k1_wg = k1_wg;
    g_dev.k2_ticks = ticks ? ticks + 16 : nullptr; g_dev.k2_wg = k2_wg;
    g_dev.lbl_ticks = ticks ? ticks + 32 : nullptr; g_dev.lbl_col = lbl_col;
    return STTM_OK;
}
#endif

int sttm_pool2d_out_side(int side, int stride, int mode) {
    if (side < 1 || stride < 1 || mode < STTM_POOL_AVERAGE || mode > STTM_POOL_BILINEAR) return fail(STTM_ERR_ARG, "bad side/stride/mode");
    if (stride == 1) return side;
    return mode == STTM_POOL_BILINEAR ? (side + stride - 1) / stride : side / stride;
}

int sttm_pool2d(const void* x, int T, int H, int W, int C, int dtype, int mode, int stride, void* out, void* stream_) {
    if (!x || !out || T < 1 || H < 1 || W < 1 || C < 1 || stride < 1) return fail(STTM_ERR_ARG, "bad pointer or shape");
    if (dtype < 0 || dtype > 2) return fail(STTM_ERR_ARG, "bad dtype");
    if (mode < STTM_POOL_AVERAGE || mode > STTM_POOL_BILINEAR) return fail(STTM_ERR_ARG, "Unexpected mm_spatial_pool_mode: %d", mode);
    if (dtype != STTM_F32 && (C & 1)) return fail(STTM_ERR_UNSUPPORTED, "16-bit inputs need an even channel count");
    const int OH = sttm_pool2d_out_side(H, stride, mode), OW = sttm_pool2d_out_side(W, stride, mode);
    if (OH < 1 || OW < 1) return fail(STTM_ERR_ARG, "pooling window larger than the grid");
    hipError_t e = sttm::launch_pool2d(x, out, T, H, W, C, OH, OW, stride, mode, dtype, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(STTM_ERR_LAUNCH, "pool2d kernel: %s", hipGetErrorString(e));
    return STTM_OK;
}

int sttm_resize_nearest(const void* x, int T, int H, int W, int C, int dtype, int OH, int OW, void* out, void* stream_) {
    if (!x || !out || T < 1 || H < 1 || W < 1 || C < 1 || OH < 1 || OW < 1) return fail(STTM_ERR_ARG, "bad pointer or shape");
    if (dtype < 0 || dtype > 2) return fail(STTM_ERR_ARG, "bad dtype");
    if (dtype != STTM_F32 && (C & 1)) return fail(STTM_ERR_UNSUPPORTED, "16-bit inputs need an even channel count");
    hipError_t e = sttm::launch_pool2d(x, out, T, H, W, C, OH, OW, 1, STTM_POOL_NEAREST, dtype, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(STTM_ERR_LAUNCH, "resize kernel: %s", hipGetErrorString(e));
    return STTM_OK;
}

static inline int dycoke_pairs(int T) { return T / 2 + (T > 4 ? (T - 4 + 3) / 4 : 0); }

size_t sttm_dycoke_workspace_bytes(int T, int P, int k) {
    if (T < 1 || P < 1 || k < 0) { fail(STTM_ERR_ARG, "bad T/P/k"); return 0; }
    const size_t np = (size_t)dycoke_pairs(T);
    return ((np * P * 4 + 255) / 256) * 256 + ((np * (size_t)(k > 0 ? k : 1) * 4 + 255) / 256) * 256 + 256;
}

int64_t sttm_dycoke_out_rows(int T, int P, int k) {
    int64_t rows = 0;
    const int n1 = T / 2;
    for (int f = 0; f < T; ++f) {
        const bool pruned = (f & 1) || ((f & 3) == 2 && f - 2 < T - 4);
        rows += pruned ? k : P;
    }
    (void)n1;
    return rows;
}

int sttm_dycoke_ttm(const void* x, int T, int P, int C, int dtype, int k, void* workspace, size_t workspace_bytes,
                    void* out, int64_t* out_idx, void* stream_) {
    if (!x || !workspace || !out || !out_idx || P < 1 || C < 1 || k < 0 || k > P) return fail(STTM_ERR_ARG, "bad pointer or shape");
    if (T < 5) return fail(STTM_ERR_ARG, "stack expects a non-empty TensorList (dycoke_ttm needs at least 5 frames)");
    if (dtype < 0 || dtype > 2) return fail(STTM_ERR_ARG, "bad dtype");
    if (dtype != STTM_F32 && (C & 1)) return fail(STTM_ERR_UNSUPPORTED, "16-bit inputs need an even channel count");
    if (P > 2048) return fail(STTM_ERR_UNSUPPORTED, "more than 2048 tokens per frame");
    if (workspace_bytes < sttm_dycoke_workspace_bytes(T, P, k)) return fail(STTM_ERR_ARG, "workspace too small");
    const size_t np = (size_t)dycoke_pairs(T);
    char* ws = reinterpret_cast<char*>(workspace);
    float* sim = reinterpret_cast<float*>(ws);
    int32_t* keep = reinterpret_cast<int32_t*>(ws + ((np * P * 4 + 255) / 256) * 256);
    hipError_t e = sttm::launch_dycoke(x, T, P, C, dtype, k, sim, keep, out, out_idx, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(STTM_ERR_LAUNCH, "dycoke kernels: %s", hipGetErrorString(e));
    return STTM_OK;
}

size_t sttm_octree_workspace_bytes(int n_cubes, int side, int C, int dtype, int root_level) {
    if (n_cubes < 1 || C < 1 || dtype < 0 || dtype > 2) { fail(STTM_ERR_ARG, "bad cubes/C/dtype"); return 0; }
    OctPlan p;
    const int L = octree_plan(n_cubes, side, C, dtype, root_level, &p);
    if (L < 0) { fail(L, "octree levels"); return 0; }
    return p.total;
}

int sttm_octree_build(const void* x, int n_cubes, int side, int C, int dtype, float threshold, int root_level,
                      void* workspace, size_t workspace_bytes, void* feat_out, int32_t* count_out, void* stream_) {
    if (!x || !workspace || !feat_out || !count_out || n_cubes < 1 || C < 1) return fail(STTM_ERR_ARG, "bad pointer or shape");
    if (dtype < 0 || dtype > 2) return fail(STTM_ERR_ARG, "bad dtype");
    if (dtype != STTM_F32 && (C & 1)) return fail(STTM_ERR_UNSUPPORTED, "16-bit inputs need an even channel count");
    if ((int64_t)n_cubes * side * side * side >= (1ll << 31)) return fail(STTM_ERR_UNSUPPORTED, "more than 2^31 tokens");
    OctPlan p;
    const int L = octree_plan(n_cubes, side, C, dtype, root_level, &p);
    if (L == STTM_ERR_INDEX) return fail(STTM_ERR_INDEX, "list index out of range (root_level %d for a cube side of %d)", root_level, side);
    if (L < 0) return fail(L, "octree levels");
    if (workspace_bytes < p.total) return fail(STTM_ERR_ARG, "workspace too small");
    char* ws = reinterpret_cast<char*>(workspace);
    sttm::OctArgs a;
    memset(&a, 0, sizeof(a));
    a.B = n_cubes; a.C = C; a.L = L;
    for (int l = 0; l < L; ++l) a.side[l] = p.side[l];
    for (int l = 0; l < L - 1; ++l) { a.feat[l] = ws + p.off_feat[l]; a.stop[l] = reinterpret_cast<uint8_t*>(ws + p.off_stop[l]); }
    a.feat[L - 1] = x;
    a.thr_lo_sq = thr_lo_sq_of(threshold);
    a.mark = reinterpret_cast<int32_t*>(ws + p.off_mark);
    a.level_of = reinterpret_cast<uint8_t*>(ws + p.off_lvl);
    a.rows = reinterpret_cast<int32_t*>(ws + p.off_rows);
    a.count_out = count_out;
    a.out = feat_out;
    const int eb = (int)elem_bytes(dtype);
    int vec = 16 / eb;
    const uintptr_t align = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(feat_out);
    while (vec > (eb == 4 ? 1 : 2) && (C % vec || align % (vec * eb))) vec >>= 1;
    hipError_t e = sttm::launch_octree(a, dtype, vec, ws + p.off_scan, p.scan_bytes, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(STTM_ERR_LAUNCH, "octree kernels: %s", hipGetErrorString(e));
    return STTM_OK;
}

int sttm_merge_dst_idx(const int32_t* pairs, int L, int N, int32_t* rep_out, void* scratch, int32_t* iters_out,
                       void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (N < 1 || L < 0 || !rep_out || !scratch || (L > 0 && !pairs)) return fail(STTM_ERR_ARG, "bad arguments");
    int32_t* rep2 = reinterpret_cast<int32_t*>(scratch);
    int32_t* emin = rep2 + N;
    hipError_t e = sttm::launch_label_edges(pairs, L, N, rep_out, rep2, emin, iters_out, stream);
    if (e != hipSuccess) return fail(STTM_ERR_LAUNCH, "label kernel: %s", hipGetErrorString(e));
    return STTM_OK;
}

}  // extern "C"
