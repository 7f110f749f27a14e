// Internal kernel-launch interfaces shared by the .hip translation units of libsttm_hip.so.
#pragma once
#include "../../include/sttm_hip.h"
#include "sttm_common.h"

namespace sttm {

// Measurement hooks (wall_clock64 stamps of single workgroups, ablation modes) exist only in the development build
// (-DSTTM_DEV, `python -m sttm_amd.build --dev` -> libsttm_hip_dev.so); the product library carries none of them.
#ifdef STTM_DEV
struct DevHooks {
    int k1_mode;              // spatial-kernel ablation: 1 = stop after the statistics, 2 = loads + pooling only (outputs invalid)
    long long* k1_ticks;      // stamps of spatial workgroup k1_wg
    int k1_wg;
    long long* k2_ticks;      // stamps of pair workgroup k2_wg
    int k2_wg;
    long long* lbl_ticks;     // stamps of the label stage of column lbl_col
    int lbl_col;
    long long* k1_span;       // [2 * workgroups] start / end stamp of EVERY spatial workgroup (null = off)
    int k5_mode;              // group-mean ablation: 1 = no member scan (every survivor treated as a singleton; outputs invalid)
};
#define STTM_DEV_TICK(hooks, field, sel, n) \
    do { if ((hooks).field && (sel) && threadIdx.x == 0) (hooks).field[n] = wall_clock64(); } while (0)
#else
#define STTM_DEV_TICK(hooks, field, sel, n) do { } while (0)
#endif

// Per-video buffers of one launch set (blockIdx.y = video): the kernels take their arguments for video 0 and shift them.
constexpr int kBatchMax = STTM_BATCH_MAX;
struct BatchPtrs {
    const void* x[kBatchMax];
    void* feat[kBatchMax];
    int32_t* npatch[kBatchMax];
    int32_t* tlbr[kBatchMax];
    size_t ws_stride;         // bytes between the workspaces of consecutive videos
};
template <typename P> __device__ __forceinline__ void shift_ptr(P*& p, size_t bytes) {
    if (p) p = reinterpret_cast<P*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(p)) + bytes);
}

// Source taps of the pooled-input spatial kernel, per axis (0 = rows, 1 = columns) and pooled coordinate: the two source rows / columns
// and their bilinear weights (average / max: rows 2o and 2o + 1, weights unused).  Built on the host with the float arithmetic of
// bilinear_tap (sttm_common.h); passed BY VALUE as a kernel argument (2 KB), so a workgroup reads its taps with scalar loads.
constexpr int kPoolMaxSide = 64;
struct PoolTable { int i0[2][kPoolMaxSide], i1[2][kPoolMaxSide]; float l0[2][kPoolMaxSide], l1[2][kPoolMaxSide]; };

struct SpatialArgs {
    const void* x;            // [T, H, W, C] memory (channels-last view of the logical [T, C, H, W])
    int64_t sT, sH, sW;       // element strides; the channel stride is 1
    int T, H, W, C;
    LevelDims dims;
    float threshold;
    double thr_lo_sq;         // lo*|lo| with lo = smallest real that rounds (fp32, RNE) to >= threshold
    int sum_mode;             // weighted_avg: sum-pool pyramid
    int n_head;               // 0 = whole-vector cosine; > 0 = per-head cosine averaged over n_head heads
    int head_lanes;           // head_dim / vec: adjacent lanes that own one head (power of two <= 64)
    int leaves_in_x;          // x is a dense [T*H*W, C] matrix: 1x1 nodes are NOT copied to S (consumers read x)
    int k1_var;               // option (A/B, tests): 1 = interior root cells also run the general body
    // upstream pooling fused into the leaf load (k_spatial_pooled): x is the UNPOOLED [T, src_h * src_w, C] token map (sT / sH / sW
    // are ITS strides), H x W the pooled grid; pool_mode = STTM_POOL_*, pool_sh / pool_sw = src / out (bilinear source scale)
    int src_h, src_w, pool_mode;
    float pool_sh, pool_sw;
    // outputs
    void* S;                  // [T*H*W, C] node features at their origin rows (input dtype)
    uint32_t* meta;           // [T*H*W] 0 = no node starts here, else (y2 << 16) | x2
    double* inrm;             // [T*H*W] 1 / (|node feature| + 1e-8) for the temporal cosine
    int* rc_list;             // [T*R][rc_stride]: count | (1x1 count << 16), then packed (y1<<24 | x1<<16 | y2<<8 | x2)
    int rc_stride;
    int32_t* lab_row;         // [T*H*W] default labels: a node's own origin row, -1 where no node starts
    int32_t* gcnt;            // [T*H*W] default group sizes: 1 at a node's origin row, else 0
    uint32_t* cgeo;           // [H*W] per leaf position: its root cell's first leaf row / column and extent, Y1 | X1<<8 | aw<<16 | ah<<24
    int32_t* counts;          // STTM_CNT_* slots (zeroed here, filled by the later kernels)
    int32_t* frame_cnt;       // [T] zeroed here for the label stage
    int32_t* bar;             // [8] zeroed here: [1] sticky overflow flag of the pair kernel, [2..3] the 64-bit N' word, [4..5] the label stage's
                              // 64-bit grid-barrier word (arrivals + per-iteration idempotency counts)
    int32_t* col_arrive;      // [R] zeroed here: pair workgroups of a column that have published their edges
#ifdef STTM_DEV
    DevHooks dev;
#endif
};
__device__ __forceinline__ void rebase(SpatialArgs& a, const BatchPtrs& bp, int v) {
    a.x = bp.x[v];
    if (v == 0) return;
    const size_t off = (size_t)v * bp.ws_stride;
    shift_ptr(a.S, off); shift_ptr(a.meta, off); shift_ptr(a.inrm, off); shift_ptr(a.rc_list, off); shift_ptr(a.lab_row, off);
    shift_ptr(a.gcnt, off); shift_ptr(a.frame_cnt, off); shift_ptr(a.bar, off); shift_ptr(a.col_arrive, off); shift_ptr(a.cgeo, off);
    a.counts += (size_t)v * STTM_CNT_SLOTS;
}
// bytes of the block-top table of the split spatial stage (256-byte multiple; the upper-cell feature table follows it)
inline size_t split_tops_bytes(int T, const LevelDims& d, int C, int elem_bytes) {
    const int ul = d.n_level - 3;
    if (ul < 1) return 0;
    return ((size_t)T * d.h[ul] * d.w[ul] * C * elem_bytes + 255) / 256 * 256;
}
inline size_t split_ufeat_bytes(int T, const LevelDims& d, int C, int elem_bytes) {
    const int ul = d.n_level - 3;
    if (ul < 1) return 0;
    return (size_t)T * d.h[0] * d.w[0] * depth_base(ul) * C * elem_bytes;
}
// tops != null: trees of 4 and more levels run in the split form (one workgroup per 3-level block + a pass over the upper levels;
// whole-vector cosine only); tops = [T][h_block_level][w_block_level][C] elements of the input dtype per video, inside the workspace
hipError_t launch_spatial(const SpatialArgs& a, const BatchPtrs& bp, int n_videos, int dtype, int vec, int nt, hipStream_t stream, void* tops = nullptr);
hipError_t launch_node_apply(const SpatialArgs& a, int dtype, int vec, int nt, hipStream_t stream);
// The column-walk spatial stage (spatial_col.inc, round 6; launch sets of several videos): one workgroup per (root cell, chunk of
// `frames` frames) that also runs the pair stage on rows kept in LDS.  The pointers are TemporalArgs' (video 0; shifted per video).
struct ColWalkArgs {
    int32_t* edges;           // [R][T-1][ecap]
    int32_t* edge_cnt;        // [R][T-1]
    int32_t* cand_cnt;        // [R][T-1]
    int ecap;
    float temporal_thresh;
    int frames;               // frames per workgroup (a chunk that does not start the clip reads one warm-up frame more)
    int cap;                  // node rows of the frame before kept in LDS (the rest is re-read from S / x)
    int pb;                   // candidate pairs per round of partial products
    int abl;                  // DEVELOPMENT build only (the product compiles it out): ablation bits for measurements, outputs invalid -- 1 = no pair phase, 2 = no row stash, 4 = no warm-up frame
};
constexpr int kColStatRow = 44;       // floats per wave of the statistics table (TreeConst<3>::NSTAT, checked in spatial_col.inc)
// LDS of a column-walk workgroup (spatial_col.inc carves it in this order): node rows | 2 x 16 inverse norms, 2 x 16 + 16 pair descriptors,
// 2 (+ 2) edge counters | partial dot products | per-wave statistics
inline size_t col_walk_lds_bytes(int nt, int cap, int pb) {
    return (size_t)cap * nt * 16 + sizeof(double) * 64 + sizeof(int) * 20 + sizeof(float) * ((size_t)pb * nt + (size_t)(nt / 64) * kColStatRow);
}
hipError_t launch_spatial_col(const SpatialArgs& a, const BatchPtrs& bp, const ColWalkArgs& cw, int n_videos, int dtype, int nt, hipStream_t stream);
// 3-level trees, 16-byte packs, whole-vector cosine: the spatial stage reading the unpooled token map (a.src_h x a.src_w, a.pool_mode)
hipError_t launch_spatial_pooled(const SpatialArgs& a, const BatchPtrs& bp, int n_videos, int dtype, int nt, hipStream_t stream);      // (pooled grid side <= kPoolMaxSide)

struct TemporalArgs {
    int T, H, W, C, R;        // R = root cells per frame
    LevelDims dims;
    int dtype, vec;
    int pair_vec;             // pack width of the pair kernel (the order its dot products are summed in; = the column walk's where that can run)
    float temporal_thresh;    // cosine threshold of the pair filter
    int temporal_on;          // 0: no temporal stage (labels stay the identity).  The merge: temporal_thresh > 0 (quadtree_builder.py:217);
                              // the stand-alone temporal stage: always (cross_frame_node_merging_fast filters with whatever threshold it is given)
    int n_head, head_lanes;   // per-head cosine in the pair filter (0 = whole vector)
    int inline_norms;         // the pair kernel computes |x| itself (spatial stage ran per-head, pair filter does not)
    int slow_ver;
    int weighted_avg;
    int max_slots;            // T * (largest root-cell area in leaves)
    int force_gmem;           // option: run the label stage on the global-memory path
    const void* S;
    const void* xrows;        // non-null: x as a dense [T*H*W, C] matrix; rows of 1x1 nodes live there, not in S
    const uint32_t* meta;
    const double* inrm;
    const int* rc_list;
    int rc_stride;
    // scratch
    int32_t* edges;           // [R][T-1][ecap] kept edges of one frame pair: (leaf offset in the earlier frame's root cell << 16) |
                              // leaf offset in the later frame's; the first edge_cnt entries of a list are valid
    int ecap;
    unsigned ecap_magic;      // ceil(2^32 / ecap): j / ecap for j < (T-1) * ecap
    int pairs_seg;            // k_pairs: consecutive frame pairs of one root cell per workgroup
    int pairs_nt;             // k_pairs block size (power of two, 64 .. 1024)
    int pairs_var;            // 9: the general pair kernel (k_pairs) also for the default shape (tests / A-B)
    float* edge_sim;          // [R][T-1][ecap] similarity of each kept edge (slow_ver only, else null)
    int32_t* edge_cnt;        // [R][T-1]
    int32_t* cand_cnt;        // [R][T-1]
    unsigned long long* col_mask;   // [R] per-column idempotency history (bit k = idempotent after iteration k+1)
    int32_t* col_arrive;      // [R] arrivals of a column's pair workgroups (zeroed by the spatial kernel)
    int32_t* frame_cnt;       // [T] survivors per frame (zeroed by the spatial kernel)
    int32_t* bar;             // [8] overflow flag, N' word, grid-barrier word (zeroed by the spatial kernel)
    int no_dense;             // option: 1 = never use the uncompacted (slot-indexed) form of the column label stage, 2 = its round-3 form
    int no_fuse, want_fold;   // options: two-launch label path / label stage inside the pair kernel (opt-in)
    int fold_kb;              // LDS budget (KB) of a pair workgroup when the label stage is folded in
    int fold_labels;          // the last pair workgroup of a column to arrive runs that column's label stage
    int fold_cap;             // ... with room for this many active nodes / kept edges in LDS (global scratch beyond)
    int label_nt;             // threads per column workgroup of the stand-alone label kernels
    int32_t* colscratch;      // label arrays of columns that do not fit LDS
    int gm_split;             // group-mean workgroups per frame
    int32_t* lab_row;         // [T*H*W] by origin row: origin row of the node's representative (-1: no node starts here)
    int32_t* gcnt;            // [T*H*W] by origin row: members of the group this node represents; 0 = not a survivor
    const uint32_t* cgeo;     // [H*W] root-cell geometry of every leaf position (written by the spatial kernel)
    int32_t* counts;
    int32_t* counts_host;     // optional host-mapped (pinned) mirror of counts, published by the label stage with slot 7 = seq
    int seq;
    unsigned long long* early_host;   // optional pinned uint64[R]: every column's (seq << 32 | flags | survivors), stored by the column itself
    // outputs
    void* feat_out;
    int32_t* npatch_out;
    int32_t* tlbr_out;
    int32_t* idx_out;         // optional [T*H*W] int32: t*H*W + y1*W + x1 of every merged token (the hook's merged_token_1d_idx); single video only
#ifdef STTM_DEV
    DevHooks dev;
#endif
};
__device__ __forceinline__ void rebase(TemporalArgs& a, const BatchPtrs& bp, int v) {
    if (a.xrows) a.xrows = bp.x[v];
    a.feat_out = bp.feat[v]; a.npatch_out = bp.npatch[v]; a.tlbr_out = bp.tlbr[v];
    if (v == 0) return;
    const size_t off = (size_t)v * bp.ws_stride;
    shift_ptr(a.S, off); shift_ptr(a.meta, off); shift_ptr(a.inrm, off); shift_ptr(a.rc_list, off);
    shift_ptr(a.edges, off); shift_ptr(a.edge_sim, off); shift_ptr(a.edge_cnt, off); shift_ptr(a.cand_cnt, off);
    shift_ptr(a.col_mask, off); shift_ptr(a.col_arrive, off); shift_ptr(a.frame_cnt, off); shift_ptr(a.bar, off);
    shift_ptr(a.colscratch, off); shift_ptr(a.lab_row, off); shift_ptr(a.gcnt, off); shift_ptr(a.cgeo, off);
    a.counts += (size_t)v * STTM_CNT_SLOTS;
    if (a.counts_host) a.counts_host += (size_t)v * STTM_CNT_SLOTS;
    if (a.early_host) a.early_host += (size_t)v * STTM_EARLY_SLOTS;
    a.seq += v;
}
// every launcher takes the arguments of video 0 plus the per-video buffers; n_videos <= kBatchMax
hipError_t launch_pairs(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream);
hipError_t launch_slow_filter(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream);
hipError_t launch_col_labels(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, bool probe, hipStream_t stream);
bool labels_can_fuse(const TemporalArgs& a, int n_videos, int concurrent_sets = 1);
bool labels_can_fold(const TemporalArgs& a, int n_videos, int* cap);
hipError_t launch_labels_fused(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream);
hipError_t launch_group_mean(const TemporalArgs& a, const BatchPtrs& bp, int n_videos, hipStream_t stream);
// stand-alone temporal stage: a caller's node list -> the layout the spatial kernel leaves behind (single video)
hipError_t launch_ingest_nodes(const TemporalArgs& a, const void* feat, const int32_t* tlbr, int n_nodes, void* S, uint32_t* meta,
                               double* inrm, int* rc_list, uint32_t* cgeo, hipStream_t stream);
size_t colscratch_ints(int T, int H, int W, int R);
void pairs_shape(int T, int R, int fold, int want_seg, int want_nt, int* seg, int* nt);

int tome_flat_mode();        // 256-tile ToMe match kernels: 1 = tile products spread evenly over the CUs when that is shorter, 0 = never, 2 = always
int tome_split_mode();       // 0 = fp32-input MFMA match kernel; > 0 = fp16 two-plane split variants (api.hip: Config)

hipError_t launch_pool2d(const void* x, void* out, int T, int H, int W, int C, int OH, int OW, int stride, int mode, int dtype,
                         hipStream_t stream);
constexpr int kOctMaxLevels = 8;

struct OctArgs {
    int B, C, L;                         // cubes, channels, pyramid levels (0 = root level, L-1 = leaves)
    int side[kOctMaxLevels];             // cells per axis
    const void* feat[kOctMaxLevels];     // [B, side^3, C]; feat[L-1] = the input
    uint8_t* stop[kOctMaxLevels];        // [B, side^3] for levels 0 .. L-2
    double thr_lo_sq;
    int32_t* mark;                       // [B * S^3] 1 = an emitted node starts at this leaf
    uint8_t* level_of;                   // [B * S^3] level of that node
    int32_t* rows;                       // [B * S^3] exclusive scan of mark
    int32_t* count_out;
    void* out;
};

size_t octree_scan_bytes(int64_t n);
hipError_t launch_octree(OctArgs& a, int dtype, int vec, void* scan_tmp, size_t scan_bytes, hipStream_t stream);
hipError_t launch_dycoke(const void* x, int T, int P, int C, int dtype, int k, float* sim, int32_t* keep, void* out, int64_t* out_idx,
                         hipStream_t stream);
hipError_t launch_label_edges(const int32_t* pairs, int L, int N, int32_t* rep_out, int32_t* rep2, int32_t* emin,
                              int32_t* iters_out, hipStream_t stream);

}  // namespace sttm
