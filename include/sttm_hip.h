/*
 * libsttm_hip.so -- C ABI of the MI355X (gfx950) implementation of STTM's token-merging hot path.
 *
 * The reference (HYUNJS/STTM) is pure Python/PyTorch and has no FFI of its own; the boundary its callers
 * see is the pair of L1 functions
 *     get_quadtree_features(...)   token_merging_utils/quadtree_interface.py:5-13
 *     get_tome_features(...)       token_merging_utils/tome_interface.py:3-9
 * called from the patched decoder forward
 *     token_merging_monkey_patch/quadtree_attn_monkey_patch.py:100-101
 *     token_merging_qwen2vl_monkey_patch/quadtree_attn_monkey_patch.py:100-101
 *     token_merging_monkey_patch/tome_attn_monkey_patch.py:99
 * The entry points below are what a ctypes binding of those two functions needs (see INTEGRATION.md for
 * the stub); `sttm_amd/quadtree_interface.py` and `sttm_amd/tome_interface.py` are that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the library never allocates device memory and never synchronises: the caller owns all buffers
 *     (sized with the *_workspace_bytes / worst-case rules below) and the stream.  ONE exception: the stage-skewed form of
 *     sttm_quadtree_merge_batch (default, "batch_streams" >= 2) runs on INTERNAL non-blocking streams -- at most 8 per host
 *     thread and device, created on first use, forked from and joined back into the caller's stream with events, so the
 *     caller's stream order is preserved --; they live until the thread calls sttm_release_streams() (or the process ends);
 *     "batch_streams" <= 1 keeps every launch on the caller's stream;
 *   - return value 0 = success, < 0 = error (sttm_last_error() returns a thread-local message);
 *   - dtype codes: 0 = float32, 1 = bfloat16, 2 = float16;
 *   - strides are in ELEMENTS of the logical [T, C, H, W] tensor the reference API receives; the
 *     production layout is a channels-last view (stride_c == 1), which is required here.
 */
#ifndef STTM_HIP_H
#define STTM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STTM_ABI_VERSION 8

#define STTM_F32 0
#define STTM_BF16 1
#define STTM_F16 2

/* error codes */
#define STTM_OK 0
#define STTM_ERR_ARG (-1)        /* bad argument (message says which)                               */
#define STTM_ERR_UNSUPPORTED (-2)/* valid for the reference, not implemented on the device path     */
#define STTM_ERR_LAUNCH (-3)     /* HIP reported a launch error                                     */
#define STTM_ERR_INDEX (-4)      /* root_level out of range (the reference raises IndexError)       */
#define STTM_ERR_PARITY (-5)     /* weighted_avg on a mixed-parity level (reference: RuntimeError)  */
#define STTM_ERR_TIMEOUT (-6)    /* sttm_wait_counts: the counts were not published in time         */

/* slots of the int32 `counts` array written by sttm_quadtree_merge (device memory, >= 8 ints) */
#define STTM_CNT_NODES 0      /* N  : nodes after the spatial stage                                  */
#define STTM_CNT_CANDIDATES 1 /* L  : cross-frame candidate pairs                                    */
#define STTM_CNT_EDGES 2      /* L' : pairs kept by the cosine filter                                */
#define STTM_CNT_OUT 3        /* N' : merged tokens written to the outputs                           */
#define STTM_CNT_ITERS 4      /* label-propagation iterations                                        */
#define STTM_CNT_OVERFLOW 5   /* != 0 : outputs invalid.  Bit 6 (STTM_OVF_BARRIER_TIMEOUT): the in-kernel grid barrier of the fused
                                 label stage timed out (other streams held the CUs) -- repeat the call with STTM_FLAG_NO_FUSE;
                                 lower bits: an internal list overflowed (never expected) */
#define STTM_OVF_BARRIER_TIMEOUT 64
#define STTM_CNT_LEAFNODES 6  /* spatial-stage nodes that are single 1x1 tokens (their rows are never copied) */
#define STTM_CNT_SLOTS 8

int sttm_abi_version(void);
const char* sttm_last_error(void);
/* Short hash of the sources this library was built from (measurements under profiles/ name the build they were taken on). */
const char* sttm_build_tag(void);

/* Number of pyramid levels the reference would build for this grid/root_level
 * (quadtree_builder.py:101-117), or a negative error code (STTM_ERR_INDEX / STTM_ERR_ARG). */
int sttm_quadtree_num_levels(int H, int W, int root_level);

/* Bytes of scratch `sttm_quadtree_merge` needs for this configuration (0 on error). */
size_t sttm_quadtree_workspace_bytes(int T, int H, int W, int C, int dtype, int root_level);

/*
 * Full STTM merge of one video: replaces quadtree_build_video (quadtree_builder.py:85-235) incl.
 * cross_frame_node_merging_fast (quadtree_temporal_merger.py:271-287).
 *
 *   x                     logical [T, C, H, W], element strides (stride_t, stride_c, stride_h, stride_w);
 *                         stride_c must be 1.  Not written.
 *   threshold             spatial cosine threshold            (reference arg `threshold`)
 *   temporal_thresh       <= 0 disables the temporal stage    (reference arg `temporal_thresh`)
 *   root_level            index into the level-size list, negatives allowed (`root_level`)
 *   weighted_avg          0/1: sum-pool pyramid + divide by patch count (`weighted_avg`)
 *   head_dim              0 = whole-vector cosine; > 0 = per-head cosine, mean over heads (`head_dim`)
 *   slow_ver              0/1: cross_frame_node_merging_slow -- per frame pair, sort kept edges by similarity and drop
 *                         adjacent duplicates of the same src (quadtree_temporal_merger.py:75-121)
 *   workspace             >= sttm_quadtree_workspace_bytes(...) bytes, 256-byte aligned
 *   feat_out              [T*H*W, C] worst case, dtype of x;  rows [0, N') are valid on return
 *   npatch_out            [T*H*W] int32
 *   tlbr_out              [T*H*W, 5] int32 (t, y1, x1, y2, x2)
 *   counts                int32[STTM_CNT_SLOTS] (device); read counts[STTM_CNT_OUT] after the stream
 *                         has drained to learn N'.
 */
int sttm_quadtree_merge(const void* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                        int T, int C, int H, int W, int dtype,
                        float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                        void* workspace, size_t workspace_bytes,
                        void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                        void* stream);

/*
 * The spatial stage alone (SURVEY Appendix E's `sttm_quadtree_spatial`): quadtree_build_video with temporal_thresh <= 0 -- per-frame
 * pyramid, split decisions and node emission (quadtree_builder.py:85-215), nodes in (t, y1, x1) order.  Same buffers as
 * sttm_quadtree_merge; counts[STTM_CNT_OUT] = number of nodes.  (The temporal stage's stand-alone entry point is sttm_temporal_merge below; inside the merge it reads the
 * node tables the spatial kernel leaves in the workspace -- root-cell node lists, inverse norms, default labels -- not a caller's
 * node list; cross_frame_node_merging_fast is never called on its own by the reference's L1, quadtree_builder.py:217-223.)
 */
int sttm_quadtree_spatial(const void* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                          int T, int C, int H, int W, int dtype, float threshold, int root_level, int weighted_avg, int head_dim,
                          void* workspace, size_t workspace_bytes, void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                          void* stream);

/* The temporal stage alone, on a caller's node list: cross_frame_node_merging_fast / _slow of the reference
 * (token_merging_utils/quadtree_temporal_merger.py:271-299; the second name SURVEY Appendix E proposed).
 *   node_feat   [n_nodes, C] contiguous rows (dtype as below), node_tlbr  int32 [n_nodes, 5] = (t, y1, x1, y2, x2), y2 / x2 exclusive
 *   T, H, W, root_level   the token grid and the root level the nodes come from.  Every box must be ONE cell of that quadtree partition
 *                         (what quadtree_build_video / sttm_quadtree_spatial emit) and the boxes of a frame must be disjoint.  A node
 *                         that is not -- outside the grid, not a cell, a duplicated origin, overlapping another node, or more nodes
 *                         than leaves in a root cell -- is left out and sets counts[STTM_CNT_OVERFLOW] (outputs invalid); the kernels
 *                         stay inside their buffers whatever the list holds.
 *   temporal_thresh       the pair filter keeps sim >= temporal_thresh, ALSO for thresholds <= 0 (quadtree_temporal_merger.py:70-71;
 *                         it is quadtree_build_video that skips the stage for temporal_thresh <= 0, not this function)
 *   head_dim              0 = whole-vector cosine; > 0 = per-head cosine averaged over heads (:65-68; ABI v7); ignored with slow_ver
 *                         like cross_frame_node_merging_slow ignores it
 * Workspace, outputs, counts and error codes as sttm_quadtree_merge (size: sttm_quadtree_workspace_bytes for the same grid).  The node
 * rows are copied once (into the origin-row layout of the merge); num_patches are the box areas; outputs are ordered by (t, y1, x1) and a
 * group is represented by its first node in that order (= the reference's lowest index when the list is sorted, as quadtree_build_video
 * passes it). */
int sttm_temporal_merge(const void* node_feat, const int32_t* node_tlbr, int n_nodes, int T, int C, int H, int W, int dtype,
                        float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                        void* workspace, size_t workspace_bytes, void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                        void* stream);

/*
 * Same merge, but the counts are ALSO published into `counts_host` -- int32[STTM_CNT_SLOTS] of pinned, device-mapped
 * host memory -- by the kernel that computes N' (the one before the feature gather), with slot STTM_CNT_SLOTS-1 set
 * to `seq` last (system-scope release).  sttm_wait_counts waits (no HIP call: a short spin, then a yielding poll) until
 * that slot equals seq: the caller learns N' -- which it needs to size the tensors it returns -- while the feature kernel
 * is still running, and returns without a stream synchronisation; consumers of the outputs are stream-ordered as usual.
 * counts_host may be NULL (then this is exactly sttm_quadtree_merge).
 *
 * events: NULL, or STTM_EVENT_SLOTS caller-created hipEvent_t handles.  The call records them on `stream` around its
 * kernels: [0] start, [1] after the spatial kernel, [2] after the pair kernel (which also runs the label stage when that is
 * folded into it), [3] after the stand-alone label kernel(s) (== [2] in time when there are none), [4] after the
 * group-mean kernel.  The caller owns the events and reads them (hipEventElapsedTime) after synchronising on [4]; the
 * library keeps no profiling state, so concurrent callers on different streams do not interfere.
 */
#define STTM_EVENT_SLOTS 5
int sttm_quadtree_merge_async(const void* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                              int T, int C, int H, int W, int dtype,
                              float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                              void* workspace, size_t workspace_bytes,
                              void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                              int32_t* counts_host, int seq, void* const* events, void* stream);
int sttm_wait_counts(const int32_t* counts_host, int seq, int timeout_us);

/*
 * The same call with its arguments in one block (ABI v5) -- a binding that issues many calls keeps the block and rewrites the
 * few fields that change (pointers, seq), instead of marshalling 27 scalars per call -- plus EARLY publication of N':
 * with `early_host` (pinned, device-mapped uint64[STTM_EARLY_SLOTS]) every root-cell column of the label stage stores ONE
 * 64-bit word  seq << 32 | flags | survivors of the column  into early_host[column] as its last action (a single naturally
 * aligned system-scope store: no fence, no returning atomic), i.e. one kernel boundary before the group-mean kernel's first
 * workgroup can publish counts_host.  The call sets n_early = number of columns that will report (0: not used -- more
 * columns than slots, or early_host NULL); sttm_wait_counts_early sums them.  Flags: bit 31 = the fused label stage's grid
 * barrier timed out (STTM_OVF_BARRIER_TIMEOUT), bit 30 = an internal list overflowed.
 */
#define STTM_EARLY_SLOTS 64
/* Per-call options (ABI v6).  STTM_FLAG_NO_FUSE: THIS call runs the label stage as two launches (no in-kernel grid barrier, no
 * residency assumption) -- what a caller does after counts[STTM_CNT_OVERFLOW] reported STTM_OVF_BARRIER_TIMEOUT, without touching
 * the process-wide "no_fuse" switch that other threads' calls read. */
#define STTM_FLAG_NO_FUSE 1
typedef struct sttm_merge_args {
    const void* x; int64_t stride_t, stride_c, stride_h, stride_w;
    int32_t T, C, H, W, dtype;
    float threshold, temporal_thresh;
    int32_t root_level, weighted_avg, head_dim, slow_ver;
    void* workspace; size_t workspace_bytes;
    void* feat_out; int32_t* npatch_out; int32_t* tlbr_out; int32_t* counts;
    int32_t* counts_host; int32_t seq;
    int32_t n_early;                 /* OUT */
    uint64_t* early_host;
    void* const* events; void* stream;
    int32_t flags;                   /* ABI v6: per-call options, STTM_FLAG_* (0 = defaults) */
    int32_t* idx_out;                /* ABI v6: NULL, or [T*H*W] int32: t*H*W + y1*W + x1 of every merged token, rows [0, N') -- the
                                        merged_token_1d_idx the patched forward derives from tlbr (quadtree_attn_monkey_patch.py:103-104),
                                        written by the group-mean kernel instead of by three elementwise launches of the caller */
} sttm_merge_args;
int sttm_quadtree_merge_packed(sttm_merge_args* args);
/* Waits until either all n_early column words carry `seq` (then out[0] = N', out[1] = overflow flags in the encoding of
 * counts[STTM_CNT_OVERFLOW]) or counts_host[STTM_CNT_SLOTS-1] == seq (then out = the published counts), like sttm_wait_counts. */
int sttm_wait_counts_early(const int32_t* counts_host, const uint64_t* early_host, int n_early, int seq, int timeout_us, int32_t* out);

/*
 * Extension (the reference API is one video per call): the same merge for `n_videos` videos of ONE shape, dtype and stride
 * set in one set of launches -- every kernel gets a second grid dimension over the videos, so the launch ramps and tails and
 * the latency-bound label stage of one video overlap the bandwidth-bound kernels of its neighbours.  Results are identical
 * to n_videos separate calls.
 *   x, feat_out, npatch_out, tlbr_out   HOST arrays of n_videos device pointers (per-video buffers as in sttm_quadtree_merge)
 *   workspace                           n_videos * workspace_stride bytes; workspace_stride >= sttm_quadtree_workspace_bytes(...)
 *                                       and a multiple of 256
 *   counts                              device int32[n_videos][STTM_CNT_SLOTS]
 *   counts_host                         NULL or pinned int32[n_videos][STTM_CNT_SLOTS]; video v publishes seq + v in its last slot (written by
 *                                       the group-mean kernel, i.e. late: with early_host the host learns N' one kernel earlier and can
 *                                       prepare its next call while the feature gather of this one is still running)
 * Any n_videos >= 1.  Round 5 (stage-skewed form, default): the videos are dealt out to "batch_streams" internal streams (created on
 * first use, per host thread and device) in launch sets of "batch_sub" videos, forked from and joined back into `stream` with events, so
 * that consecutive videos sit in DIFFERENT stages at any moment; batch_streams <= 1 selects the lockstep form (every kernel once per
 * group of STTM_BATCH_MAX videos on `stream` itself).  Both forms run the same kernels with the same per-video arguments.
 */
#define STTM_BATCH_MAX 16
int sttm_quadtree_merge_batch(int n_videos, const void* const* x, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                              int T, int C, int H, int W, int dtype,
                              float threshold, float temporal_thresh, int root_level, int weighted_avg, int head_dim, int slow_ver,
                              void* workspace, size_t workspace_stride,
                              void* const* feat_out, int32_t* const* npatch_out, int32_t* const* tlbr_out, int32_t* counts,
                              int32_t* counts_host, int seq, void* const* events, void* stream, int flags /* STTM_FLAG_* */,
                              uint64_t* early_host /* NULL, or pinned uint64[n_videos][STTM_EARLY_SLOTS]: video v's label-stage columns store their
                                                      words (seq + v) << 32 | flags | survivors there, as in sttm_quadtree_merge_packed */,
                              int* n_early_out /* columns per video that report (0: not used); wait with sttm_wait_counts_early per video */);

/*
 * Tuning and test switches (process-wide; none of them changes results, except "tome_split" within the fp32 rounding noise of
 * the ToMe scores).  Defaults come from the environment variable
 * STTM_<KEY> read once at first use; sttm_configure overrides a key at run time (call it while no merge is being issued from
 * another thread).  Returns STTM_ERR_ARG for an unknown key.  Keys:
 *   "pairs_seg"   consecutive frame pairs of one root cell per pair workgroup (0 = automatic: 1; with fold_labels about one
 *                 workgroup per CU)
 *   "pairs_nt"    pair-kernel block size (0 = automatic: 256 for one pair, 128 threads per pair of a run, up to 1024)
 *   "gm_split"    group-mean workgroups per frame (0 = automatic: ~4096 / T for one video, ~1024 / T in launch sets of several videos)
 *   "label_nt"    threads per column of the stand-alone label kernels (256 / 512 / 1024; 0 = automatic: 1024 for one video, 256 in launch
 *                 sets of several videos)
 *   "vec16" / "vec32"   force the pack width of 16-bit / 32-bit inputs in the spatial kernel (0 = automatic)
 *   "fold_kb"     LDS budget (KB) per pair workgroup when the label stage runs inside the pair kernel (default 64)
 *   "fold_labels" 1: run a column's label stage inside the pair kernel, in the column's last workgroup to finish (default 0:
 *                 stand-alone label kernel; the folded form is no faster on MI355X and is kept for experiments)
 *   "pairs_var"   9: the general pair kernel instead of the lean 256-thread form (tests / A-B)
 *   "k1_var"      development build only (A/B): 1 = 3- and 4-level trees run the general spatial body (the one deeper trees use)
 *   "k1_split"    spatial stage of trees with 4 and more levels: 0 (default) one workgroup per 3-level block + a pass over the upper
 *                 levels (two launches; whole-vector cosine); -1 = one workgroup per root cell (the form the per-head cosine always
 *                 uses; 6-level trees then need <= 768 lanes per token row); 5 = the split form from 5 levels on.  Same outputs.
 *   "no_dense"    1: column label stage always on compact ids (default: columns of <= 3072 slots work on their slots directly);
 *                 2: the round-3 form of the slot-indexed stage (three barriers per iteration; A/B)
 *   "no_fuse"     1: stand-alone label stage as two launches (no in-kernel grid barrier)
 *   "force_gmem_labels"  1: label stage on global scratch instead of LDS (the path of columns too large for LDS)
 *   "batch_streams" / "batch_sub"   sttm_quadtree_merge_batch: internal streams (default 3; <= 1 = lockstep form) and videos per launch
 *                 set on a stream (default 8, at most STTM_BATCH_MAX)
 *   "col_walk"    the column-walk spatial stage (csrc/spatial_col.inc: one workgroup per root-cell column and chunk of "col_frames"
 *                 frames that also runs the pair stage on node rows kept in LDS; 3-level trees, whole-vector cosine, fast pair filter):
 *                 0 (default) never -- it moves 17 % fewer bytes and is slower --, 1 launch sets of several videos,
 *                 2 also one-video calls (tests).  "col_frames" (8), "col_cap" / "col_pb" (0 = what the LDS budget allows) shape it.
 *                 Bit-identical outputs in every setting.
 *   "pair_vec"    pack width of the pair kernel = the order its dot products are summed in: 0 (default) the column walk's (the spatial
 *                 kernel's 16-byte packs) wherever that stage could run, so that both forms give the same bits; -1 the row kernels'
 *                 width (rounds 1-5: 8 floats); differences are confined to the last bits of a similarity that sits on the threshold.
 *   "tome_split"  ToMe match kernel for float32 inputs: unit rows as two fp16 planes (h + l of 4096 v, residual <= 2^-23), scores from
 *                 products of the planes on the fp16 matrix pipe with fp32 accumulation (each product exact).  2 (default since round 5):
 *                 l.h + h.l + h.h -- error bound 4.8e-7 + the fp32 accumulation that every sgemm has; measured against float64 on the
 *                 128-frame clip 9.1e-7, the same as with four terms and below the fp32-input MFMA kernel's 1.4e-6; 20 % fewer MFMAs;
 *                 1: the same plus l.l (bound 2.4e-7 + accumulation); kernel picked by size in both; 0: fp32-input MFMA
 *                 (v_mfma_f32_32x32x2_f32), 2.9x slower; 3 / 4: force the 128-tile / 256-tile kernel with four terms, 5 / 6 with three (tests);
 *                 7: the FOUR-WAVE form of the 256-tile kernel, three terms, also for 16-bit inputs (128 x 128 wave tiles, one wave per SIMD,
 *                 the 256 accumulators in literally named AGPRs; round 6 -- no faster than the eight-wave form)
 *   "tome_flat"   256-tile ToMe match kernels: 1 (default) spread all tile products evenly over one workgroup per CU when that
 *                 shortens the per-CU critical path against the best per-a-tile split (69 x 69 tiles at T = 180: 19 instead of 23
 *                 products per workgroup), 0 never, 2 always.  Same scores, same first-maximum argmax.
 */
int sttm_configure(const char* key, int value);

/* ABI v8: destroys the internal streams / events sttm_quadtree_merge_batch created for the CALLING host thread (all devices), after
 * synchronising them; returns how many streams were released.  The next batch call creates them again.  Call it from a thread that
 * is done with the library (thread-local state is not reachable from other threads). */
int sttm_release_streams(void);

/*
 * Position-embedding ablation (pos_embs argument of get_quadtree_features; quadtree_spatial_merger.py:88-153,
 * quadtree_temporal_merger.py:153-169): pool a side tensor `v` (logical [T, Cv, H, W], channels-last view) over the
 * nodes and groups of the sttm_quadtree_merge call that ran LAST on this stream with this workspace (same T/H/W/
 * root_level; C_feat / dtype_feat are that call's feature width and dtype, needed to find the buffers again).
 * sum_mode = pos_emb_weighted_avg (sum-pool + divide by patches).  out: [T*H*W, Cv] worst case, rows [0, N') valid.
 */
int sttm_quadtree_apply(const void* v, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                        int T, int Cv, int H, int W, int dtype_v, int sum_mode,
                        int C_feat, int dtype_feat, int root_level, void* workspace, size_t workspace_bytes,
                        const int32_t* counts, void* out, void* stream);

/*
 * Label propagation on an explicit edge list: replaces get_merge_dst_idx_safe
 * (quadtree_temporal_merger.py:223-269).  pairs is [L, 2] int32 (dst, src); rep_out is [N] int32;
 * scratch is >= (N + L) * 4 bytes; iters_out (device int32, may be NULL) receives the iteration count.
 */
int sttm_merge_dst_idx(const int32_t* pairs, int L, int N, int32_t* rep_out, void* scratch, int32_t* iters_out,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * ToMe baseline: ONE iteration of tome_per_video's loop (tome_token_merger.py:143-149), i.e.
 * bipartite_soft_matching (:13-57) + merge_wavg (:77-91) on a [n, C] token matrix.  dtype float32 (fp32 scores from exact
 * products: two-plane fp16 split on the fp16 MFMA, or the fp32-input MFMA -- see "tome_split"), bfloat16 or float16 (every
 * intermediate rounded to the input dtype like the reference's torch ops on 16-bit hidden states; bf16 / f16 MFMA; x_out has the input dtype, size / size_out stay float32 arrays holding
 * dtype-representable values).
 *
 *   x          [n, C] row-major tokens          size  [n] token sizes, or NULL for all ones (first iteration)
 *   idx        [n] int64 token ids, or NULL for 0..n-1 (first iteration)    r     tokens to remove, 1 <= r <= n / 2 (callers clamp like :26)
 *   x_out      [n - r, C]   size_out [n - r]    idx_out [n - r]
 *              rows: the (ceil(n/2) - r) unmerged even tokens in descending best-score order, then every odd token
 *              in original order with its merged sources folded in (size-weighted average, sources added in rank order)
 *   node_max_out / node_idx_out   optional [ceil(n/2)] copies of scores.max(-1) (:36) for inspection, may be NULL
 * Enqueued on `stream` without any host synchronisation; workspace >= sttm_tome_workspace_bytes(n, C, n_head).
 * Limit: the unit-row matrix of one token half, ceil(n/2) x (C / n_head rounded up to 64) x 4 bytes, must stay below 2 GiB (fp32
 * C = 1024: a million tokens); beyond it sttm_tome_workspace_bytes returns 0 and sttm_tome_step STTM_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------------ */
size_t sttm_tome_workspace_bytes(int n, int C, int n_head);
int sttm_tome_step(const void* x, const float* size, const int64_t* idx, int n, int C, int n_head, int r, int dtype,
                   void* workspace, size_t workspace_bytes, void* x_out, float* size_out, int64_t* idx_out,
                   float* node_max_out, int32_t* node_idx_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Upstream 2-D pooling of the projected vision tokens: replaces LlavaMetaForCausalLM.get_2dPool
 * (llava/model/llava_arch.py:173-198): x [T, H*W, C] row-major tokens -> out [T, OH*OW, C], same dtype.
 *   mode STTM_POOL_AVERAGE / STTM_POOL_MAX : stride x stride windows, OH = H / stride (floor)   (:185-188)
 *   mode STTM_POOL_BILINEAR                : F.interpolate(size = ceil(H / stride)), align_corners = False (:189-192)
 * sttm_pool2d_out_side gives OH (or OW) for one side; stride == 1 is the caller's identity case (:174-175).
 * Enqueued on `stream`, no workspace, no host synchronisation.
 * ------------------------------------------------------------------------------------------------ */
#define STTM_POOL_AVERAGE 0
#define STTM_POOL_MAX 1
#define STTM_POOL_BILINEAR 2
#define STTM_POOL_NEAREST 3   /* only through sttm_resize_nearest */
int sttm_pool2d_out_side(int side, int stride, int mode);
int sttm_pool2d(const void* x, int T, int H, int W, int C, int dtype, int mode, int stride, void* out, void* stream);

/*
 * The merge reading the UNPOOLED token map: get_2dPool (llava/model/llava_arch.py:173-198) fused into the leaf
 * load of the spatial kernel -- every leaf of the H x W quadtree grid is built from its four source tokens on the fly, so the pooled
 * map is neither written nor re-read.  Same results as  sttm_pool2d(x_tokens -> pooled)  followed by  sttm_quadtree_merge(pooled).
 *   x_tokens   [T, src_h * src_w, C] row-major (e.g. 27 x 27 = 729 projected SigLIP tokens per frame, video_feat_llavavideo.py:89-95)
 *   pool_mode / pool_stride   as sttm_pool2d: STTM_POOL_BILINEAR with any stride >= 2, STTM_POOL_AVERAGE / STTM_POOL_MAX with stride 2
 *   H = W = sttm_pool2d_out_side(src, pool_stride, pool_mode) is the quadtree's grid: workspace >= sttm_quadtree_workspace_bytes(T, H, W, C,
 *   dtype, root_level), outputs [T*H*W, ...] worst case as in sttm_quadtree_merge; counts_host / seq as in sttm_quadtree_merge_async.
 * STTM_ERR_UNSUPPORTED (pool with sttm_pool2d, then call sttm_quadtree_merge): trees that are not 3 levels deep for this grid / root_level
 * (every run_vidqa.sh preset is: 27 -> 14 x 14, root_level 1), rows that are not 16-byte aligned multiples of 16 bytes, other strides of
 * the average / max pool.  Whole-vector cosine only (the LLaVA-Video hook that follows get_2dPool passes head_dim only with sim_per_head).
 */
int sttm_quadtree_merge_pooled(const void* x_tokens, int T, int src_h, int src_w, int C, int dtype, int pool_mode, int pool_stride,
                               float threshold, float temporal_thresh, int root_level, int weighted_avg, int slow_ver,
                               void* workspace, size_t workspace_bytes,
                               void* feat_out, int32_t* npatch_out, int32_t* tlbr_out, int32_t* counts,
                               int32_t* counts_host, int seq, void* stream, int flags /* ABI v8: STTM_FLAG_*, as sttm_quadtree_merge_packed */);

/*
 * Nearest-neighbour resize of every frame to OH x OW: the "pyrd" baseline's F.interpolate(video, size=(s, s))
 * (token_merging_monkey_patch/pyrd_attn_monkey_patch.py:99-102; default mode "nearest").
 * x [T, H*W, C] -> out [T, OH*OW, C], same dtype; a pure row gather (exact).
 */
int sttm_resize_nearest(const void* x, int T, int H, int W, int C, int dtype, int OH, int OW, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DyCoke stage-1 pruning: replaces dycoke_ttm (token_merging_utils/dycoke_merger.py:8-83).  float32, bfloat16, float16
 * (16-bit inputs: the cosine follows the reference's per-op rounding to the input dtype).
 *   x [T*P, C] row-major tokens (P tokens per frame), k = int((1 - prune_ratio) * P) tokens kept per pruned frame.
 *   Pass 1: frames (2j, 2j+1) -- frame 2j+1 keeps its k least similar tokens (per-token cosine, ascending order).
 *   Pass 2: frames (4j, 4j+2), 4j < T-4 -- frame 4j+2 keeps its k least similar tokens w.r.t. frame 4j.
 *   out [rows, C] / out_idx [rows] int64 flat token ids, rows = sttm_dycoke_out_rows(T, P, k) (known in advance).
 * T < 5 is an argument error (the reference's torch.stack of an empty list raises there too).
 * Ties between equal similarities go to the smaller token id (torch.topk leaves that order unspecified).
 * Enqueued on `stream` without any host synchronisation; workspace >= sttm_dycoke_workspace_bytes(T, P, k).
 * ------------------------------------------------------------------------------------------------ */
size_t sttm_dycoke_workspace_bytes(int T, int P, int k);
int64_t sttm_dycoke_out_rows(int T, int P, int k);
int sttm_dycoke_ttm(const void* x, int T, int P, int C, int dtype, int k, void* workspace, size_t workspace_bytes,
                    void* out, int64_t* out_idx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Octree baseline over the whole cubes of a clip: replaces octree_build (token_merging_utils/octree_utils.py:293-373).
 *   x        [n_cubes * side, side, side, C] dense channels-last tokens (a cube = `side` consecutive frames of side x side)
 *   feat_out [n_cubes * side^3, C] worst case; rows [0, N') are valid, ordered by the first-corner leaf index (:369-373)
 *   count_out device int32: N'.   root_level indexes [2, ..., side] (halving with ceil, :312-316), negative from the end.
 * The frames that do not fill a cube are the caller's job (the reference merges them per frame with the spatial
 * quadtree, :375-378 -- sttm_quadtree_merge with temporal_thresh <= 0).
 * Returns STTM_ERR_INDEX for a root level outside the list.  No host synchronisation.
 * ------------------------------------------------------------------------------------------------ */
size_t sttm_octree_workspace_bytes(int n_cubes, int side, int C, int dtype, int root_level);
int sttm_octree_build(const void* x, int n_cubes, int side, int C, int dtype, float threshold, int root_level,
                      void* workspace, size_t workspace_bytes, void* feat_out, int32_t* count_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STTM_HIP_H */
