// Internal kernel-launch interfaces shared by the .hip translation units of libsttm_hip.so.
#pragma once
#include "../../include/sttm_hip.h"
#include "sttm_common.h"

namespace sttm {

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
    int pipeline;             // opt-in: persistent software-pipelined 3-level kernel (measured slower on MI355X)
    int dbg_mode;             // ablation (sttm_debug_spatial_ms only): 1 = stop after the statistics, 2 = loads + pooling only
    long long* dbg_ticks;     // measurement tool only (STTM_K1_TICKS=1): wall_clock64 stamps of workgroup dbg_wg, else null
    int dbg_wg;
    int leaves_in_x;          // x is a dense [T*H*W, C] matrix: 1x1 nodes are NOT copied to S (consumers read x)
    // outputs
    void* S;                  // [T*H*W, C] node features at their origin rows (input dtype)
    uint32_t* meta;           // [T*H*W] 0 = no node starts here, else (y2 << 16) | x2
    double* inrm;             // [T*H*W] 1 / (|node feature| + 1e-8) for the temporal cosine
    int* rc_list;             // [T*R][rc_stride]: count, then packed (y1<<24 | x1<<16 | y2<<8 | x2)
    int rc_stride;
    int32_t* counts;          // STTM_CNT_* slots (zeroed here, filled by the later kernels)
    int32_t* frame_cnt;       // [T] zeroed here for the label kernels
    int32_t* bar;             // [4] zeroed here: [0] grid-barrier counter of the fused label kernel, [2..3] the 64-bit arrival/total word
};
hipError_t launch_spatial(const SpatialArgs& a, int dtype, int vec, int nt, hipStream_t stream);
hipError_t launch_node_apply(const SpatialArgs& a, int dtype, int vec, int nt, hipStream_t stream);

struct TemporalArgs {
    int T, H, W, C, R;        // R = root cells per frame
    LevelDims dims;
    int dtype, vec;
    float temporal_thresh;    // <= 0: no edges (labels stay the identity)
    int n_head, head_lanes;   // per-head cosine in the pair filter (0 = whole vector)
    int inline_norms;         // the pair kernel computes |x| itself (spatial stage ran per-head, pair filter does not)
    int slow_ver;
    int weighted_avg;
    int max_slots;            // T * (largest root-cell area in leaves)
    int force_gmem;           // debug/test: run the label kernels on the global-memory path
    const void* S;
    const void* xrows;        // non-null: x as a dense [T*H*W, C] matrix; rows of 1x1 nodes live there, not in S
    const uint32_t* meta;
    const double* inrm;
    const int* rc_list;
    int rc_stride;
    // scratch
    int32_t* edges;           // [R][T-1][ecap] kept edges, packed column-local slots (dst << 16 | src), dst = earlier frame
    int ecap;
    int pairs_seg;            // k_pairs workgroup map: frames per XCD-local segment (0 = plain t-major order)
    float* edge_sim;          // [R][T-1][ecap] similarity of each kept edge (slow_ver only, else null)
    int32_t* edge_cnt;        // [R][T-1]
    int32_t* cand_cnt;        // [R][T-1]
    unsigned long long* col_mask;   // [R] per-column idempotency history (bit k = idempotent after iteration k+1)
    int32_t* frame_cnt;       // [T] survivors per frame (zeroed by the spatial kernel)
    int32_t* bar;             // [2] grid-barrier counters of the fused label kernel (zeroed by the spatial kernel)
    int no_fuse;              // debug/test: use the three-kernel label path
    int dbg_wg;               // debug: which workgroup stamps
    long long* dbg_ticks;     // debug: wall_clock64 stamps of workgroup 0 at phase boundaries (null = off)
    long long* dbg_ticks_k2;  // debug (STTM_K2_TICKS=1): stamps of pair-kernel workgroup dbg_wg_k2
    int dbg_wg_k2;
    int32_t* colscratch;      // [5*T*H*W] label arrays of columns that do not fit LDS
    int gm_split;             // group-mean workgroups per frame
    int32_t* grp_np;          // [T*H*W] by origin row: patches covered by the group
    int32_t* grp_cnt;         // [T*H*W] by origin row; 0 = not a survivor
    int32_t* grp_off;         // [T*H*W] by origin row
    int32_t* members;         // [T*H*W] origin rows (| leaf bit), grouped, ascending inside a group
    int32_t* counts;
    int32_t* counts_host;     // optional host-mapped (pinned) mirror of counts, published by the label kernel with slot 7 = seq
    int seq;
    // outputs
    void* feat_out;
    int32_t* npatch_out;
    int32_t* tlbr_out;
};
hipError_t launch_pairs(const TemporalArgs& a, hipStream_t stream);
hipError_t launch_slow_filter(const TemporalArgs& a, hipStream_t stream);
hipError_t launch_col_labels(const TemporalArgs& a, bool probe, hipStream_t stream);
bool labels_can_fuse(const TemporalArgs& a);
hipError_t launch_labels_fused(const TemporalArgs& a, hipStream_t stream);
hipError_t launch_group_mean(const TemporalArgs& a, hipStream_t stream);
bool col_labels_use_gmem(const TemporalArgs& a);

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
hipError_t launch_dycoke(const float* x, int T, int P, int C, int k, float* sim, int32_t* keep, float* out, int64_t* out_idx,
                         hipStream_t stream);
hipError_t launch_label_edges(const int32_t* pairs, int L, int N, int32_t* rep_out, int32_t* rep2, int32_t* emin,
                              int32_t* iters_out, hipStream_t stream);

}  // namespace sttm
