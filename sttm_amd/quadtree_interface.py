"""L1 function boundary of STTM, MI355X edition.

`get_quadtree_features` keeps the signature, argument meaning, return types and error behaviour of the
reference's `token_merging_utils/quadtree_interface.py:5-13` (which forwards to
`quadtree_builder.py:85-235`), but runs entirely in hand-written HIP kernels behind the C ABI of
`libsttm_hip.so`.  PyTorch is used for device memory and the current stream only.
"""
import torch

from . import _lib

_DTYPE_CODE = {torch.float32: _lib.STTM_F32, torch.bfloat16: _lib.STTM_BF16, torch.float16: _lib.STTM_F16}
_pinned_counts = _lib.BoundedCache(16)
_ws_cache = _lib.BoundedCache(8)      # (device, stream) -> scratch; bounded, see _lib.BoundedCache
_ws_bytes_cache = {}    # (T, H, W, C, dtype, root_level) -> bytes


_seq = [0]


def _counts_host(device, stream_handle):
    """Pinned landing pad of the label kernel's counts: one per (device, stream), so callers on different streams
    (e.g. two host threads) never share one."""
    key = (device, stream_handle)
    buf = _pinned_counts.get(key)
    if buf is None:
        buf = torch.zeros(_lib.CNT_SLOTS, dtype=torch.int32).pin_memory()
        _pinned_counts[key] = buf
    return buf


def quadtree_merge_raw(x, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver=False, return_ctx=False,
                       feat_dest=None):
    """Launch the merge of one video and return the worst-case-sized outputs plus the host counts.
    x: logical [T, C, H, W] CUDA tensor (any strides; channels-last views run zero-copy)."""
    if not x.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; "
                           "there is no CPU fallback")
    if x.dim() != 4:
        raise ValueError("expected a [T, C, H, W] tensor")
    if x.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {x.dtype} is not supported (float32, bfloat16, float16)")
    lib = _lib.load()
    T, C, H, W = x.shape
    if x.stride(1) != 1 or (x.data_ptr() % 16) != 0:
        # not the production (channels-last view) layout: one transposing copy
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    dev = x.device
    dtype = _DTYPE_CODE[x.dtype]
    head_dim = 0 if head_dim is None else int(head_dim)
    key = (T, H, W, C, dtype, int(root_level))
    nbytes = _ws_bytes_cache.get(key)
    if nbytes is None:
        nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, dtype, int(root_level))
        if nbytes == 0:
            code = lib.sttm_quadtree_num_levels(H, W, int(root_level))
            _lib.raise_for(code if code < 0 else _lib.ERR_ARG)
        _ws_bytes_cache[key] = nbytes
    N = T * H * W
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        skey = (dev, stream.cuda_stream)           # scratch is reused call after call on the SAME stream (stream-ordered)
        cached = _ws_cache.get(skey)
        if cached is None or cached[0].numel() < nbytes:
            cached = (torch.empty(nbytes, dtype=torch.uint8, device=dev),
                      torch.empty(_lib.CNT_SLOTS, dtype=torch.int32, device=dev))
            _ws_cache[skey] = cached
        ws, counts = cached
        if feat_dest is None:
            feat = torch.empty((N, C), dtype=x.dtype, device=dev)
        else:
            # caller-owned destination (fused slice -> merge -> concat): rows [0, N') of it receive the merged features
            feat = feat_dest
            if (feat.dim() != 2 or feat.size(1) != C or feat.dtype != x.dtype or feat.device != dev
                    or not feat.is_contiguous() or feat.data_ptr() % 16):
                raise ValueError("feat_dest must be a contiguous, 16-byte aligned [rows, C] tensor of the input's dtype and device")
        npatch = torch.empty(N, dtype=torch.int32, device=dev)
        tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev)
        host = _counts_host(dev, stream.cuda_stream)
        _seq[0] = (_seq[0] % 0x3fffffff) + 1
        seq = _seq[0]
        rc = lib.sttm_quadtree_merge_async(
            x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, dtype,
            float(threshold), float(temporal_thresh), int(root_level), int(bool(weighted_avg)), head_dim,
            int(bool(slow_ver)), ws.data_ptr(), ws.numel(), feat.data_ptr(), npatch.data_ptr(), tlbr.data_ptr(), counts.data_ptr(),
            host.data_ptr(), seq, stream.cuda_stream)
        _lib.raise_for(rc)
        # Output sizes are data dependent, so the host must learn N' -- but only N': the label kernel publishes the
        # counts into pinned memory and we spin on that, returning while the feature gather is still running.
        if lib.sttm_wait_counts(host.data_ptr(), seq, 2_000_000) != 0:
            host.copy_(counts, non_blocking=True)  # fallback: classic D2H + stream sync
            stream.synchronize()
    cnt = host.tolist()
    if cnt[_lib.CNT_OVERFLOW]:
        raise RuntimeError("libsttm_hip: internal list overflow (please report): counts=%s" % cnt)
    if return_ctx:
        return feat, npatch, tlbr, cnt, (x, ws, counts, dtype, stream)
    return feat, npatch, tlbr, cnt


def _apply_side_tensor(v, ctx, root_level, sum_mode):
    """Pool a side tensor (RoPE cos or sin, logical [T, Cv, H, W]) over the nodes / groups of the merge that just ran."""
    x, ws, counts, dtype_feat, stream = ctx
    lib = _lib.load()
    if v.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {v.dtype} is not supported")
    T, Cv, H, W = v.shape
    if v.stride(1) != 1 or (v.data_ptr() % 16) != 0:
        v = v.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out = torch.empty((T * H * W, Cv), dtype=v.dtype, device=v.device)
    rc = lib.sttm_quadtree_apply(v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), v.stride(3), T, Cv, H, W,
                                 _DTYPE_CODE[v.dtype], int(bool(sum_mode)), x.shape[1], dtype_feat, int(root_level),
                                 ws.data_ptr(), ws.numel(), counts.data_ptr(), out.data_ptr(), stream.cuda_stream)
    _lib.raise_for(rc)
    return out


_side_streams = {}


_BATCH_TIMING = []          # measurement hook (tools/batch_streams.py): non-empty list -> timestamps after the enqueue loop


def get_quadtree_features_batch(videos, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                                slow_ver=False, head_dim=None, n_streams=3):
    """Extension (not in the reference, whose API is one video per call): merge a LIST of videos and return the list of
    (features, num_patches, tlbr) triples -- results identical to calling get_quadtree_features on each.

    Videos are independent, so consecutive videos are issued round-robin on `n_streams` side streams: the latency-bound
    label kernel of one video (16 workgroups) overlaps the bandwidth-bound kernels of the next.  The host waits for the
    per-video token counts only after everything has been enqueued; the caller's current stream is made to wait for the
    side streams, so downstream ops stay stream-ordered."""
    if not videos:
        return []
    lib = _lib.load()
    dev = videos[0].device
    if not all(v.is_cuda and v.device == dev for v in videos):
        raise RuntimeError("sttm_amd runs on the GPU only: every video must be a CUDA (ROCm) tensor on one device; "
                           "there is no CPU fallback")
    head = 0 if head_dim is None else int(head_dim)
    streams = _side_streams.setdefault((dev, n_streams), None)
    with torch.cuda.device(dev):
        if streams is None:
            streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
            _side_streams[(dev, n_streams)] = streams
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        pend = []
        hosts = _pinned_counts.get((dev, "batch"))
        if hosts is None or hosts.shape[0] < len(videos):
            hosts = torch.zeros((max(64, len(videos)), _lib.CNT_SLOTS), dtype=torch.int32).pin_memory()
            _pinned_counts[(dev, "batch")] = hosts
        for j, x in enumerate(videos):
            if x.dim() != 4 or x.dtype not in _DTYPE_CODE:
                raise ValueError("expected [T, C, H, W] float32 / bfloat16 / float16 tensors")
            if x.stride(1) != 1 or (x.data_ptr() % 16) != 0:
                x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            T, C, H, W = x.shape
            dtype = _DTYPE_CODE[x.dtype]
            key = (T, H, W, C, dtype, int(root_level))
            nbytes = _ws_bytes_cache.get(key)
            if nbytes is None:
                nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, dtype, int(root_level))
                if nbytes == 0:
                    code = lib.sttm_quadtree_num_levels(H, W, int(root_level))
                    _lib.raise_for(code if code < 0 else _lib.ERR_ARG)
                _ws_bytes_cache[key] = nbytes
            st = streams[j % n_streams]
            if j < n_streams:
                st.wait_event(ready)                       # inputs were produced on the caller's stream
            skey = (dev, st.cuda_stream)
            cached = _ws_cache.get(skey)
            if cached is None or cached[0].numel() < nbytes:
                cached = (torch.empty(nbytes, dtype=torch.uint8, device=dev),
                          torch.empty(_lib.CNT_SLOTS, dtype=torch.int32, device=dev))
                for t_ in cached:                      # allocated on the caller's stream, used on the side stream
                    t_.record_stream(st)
                _ws_cache[skey] = cached
            ws, counts = cached
            N = T * H * W
            feat = torch.empty((N, C), dtype=x.dtype, device=dev)
            npatch = torch.empty(N, dtype=torch.int32, device=dev)
            tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev)
            for t_ in (feat, npatch, tlbr):
                t_.record_stream(st)
            _seq[0] = (_seq[0] % 0x3fffffff) + 1
            seq = _seq[0]
            rc = lib.sttm_quadtree_merge_async(
                x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, dtype,
                float(threshold), float(temporal_thresh), int(root_level), int(bool(weighted_avg)), head, int(bool(slow_ver)),
                ws.data_ptr(), ws.numel(), feat.data_ptr(), npatch.data_ptr(), tlbr.data_ptr(), counts.data_ptr(),
                hosts[j].data_ptr(), seq, st.cuda_stream)
            _lib.raise_for(rc)
            pend.append((feat, npatch, tlbr, seq, st, counts))
        if _BATCH_TIMING:
            import time as _t
            _BATCH_TIMING.append(_t.perf_counter())
        out = []
        for j, (feat, npatch, tlbr, seq, st, counts) in enumerate(pend):
            if lib.sttm_wait_counts(hosts[j].data_ptr(), seq, 2_000_000) != 0:
                st.synchronize()
                raise RuntimeError("libsttm_hip: the token counts of a batched merge did not arrive")
            cnt = hosts[j].tolist()
            if cnt[_lib.CNT_OVERFLOW]:
                raise RuntimeError("libsttm_hip: internal list overflow (please report): counts=%s" % cnt)
            n = cnt[_lib.CNT_OUT]
            out.append((feat[:n], npatch[:n], tlbr[:n]))
        for st in streams:
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)
    return out


def get_quadtree_features(_video_feature, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                          vis_flag=False, slow_ver=False, head_dim=None, pos_embs=None, pos_emb_weighted_avg=False):
    """Drop-in for the reference's get_quadtree_features.

    Returns (features [N', C] in the input dtype, num_patches [N'] int32, tlbr [N', 5] int32), all on the
    input's device; rows are ordered by (t, y1, x1) like the reference's output.
    """
    if vis_flag:
        raise NotImplementedError("vis_flag=True (quadtree_builder_vis) is a plotting aid and is out of scope")
    if pos_embs is None:
        feat, npatch, tlbr, cnt = quadtree_merge_raw(_video_feature, threshold, temporal_thresh, root_level,
                                                     weighted_avg, head_dim, slow_ver)
        n = cnt[_lib.CNT_OUT]
        return feat[:n], npatch[:n], tlbr[:n]
    # position-embedding ablation (quadtree_attn_monkey_patch_for_abl_pos.py): cos / sin ride along the same tree
    cos, sin = pos_embs
    if temporal_thresh <= 0 and not pos_emb_weighted_avg:
        # quirk Q11 of the reference: `pos_embs_cos` is only assigned on the temporal or the weighted path
        raise UnboundLocalError("local variable 'pos_embs_cos' referenced before assignment (reference behaviour for "
                                "pos_embs with temporal_thresh <= 0 and pos_emb_weighted_avg=False)")
    feat, npatch, tlbr, cnt, ctx = quadtree_merge_raw(_video_feature, threshold, temporal_thresh, root_level,
                                                      weighted_avg, head_dim, slow_ver, return_ctx=True)
    n = cnt[_lib.CNT_OUT]
    out_cos = _apply_side_tensor(cos, ctx, root_level, pos_emb_weighted_avg)
    out_sin = _apply_side_tensor(sin, ctx, root_level, pos_emb_weighted_avg)
    return feat[:n], npatch[:n], tlbr[:n], (out_cos[:n], out_sin[:n])



def get_quadtree_features_into(dest, _video_feature, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                               slow_ver=False, head_dim=None):
    """get_quadtree_features with a caller-owned destination: the merged features are written to dest[:N'] directly
    (dest: contiguous [rows >= T*H*W, C] tensor, e.g. a row window of the NEW hidden-state buffer), so the caller's
    `torch.cat([system, merged, instruction])` (quadtree_attn_monkey_patch.py:105) needs no copy of the merged rows.
    Nothing of `dest` beyond row N' is written.  dest must not overlap the input.
    Returns (dest[:N'], num_patches [N'], tlbr [N', 5])."""
    T, C, H, W = _video_feature.shape
    if dest.dim() != 2 or dest.size(0) < T * H * W:
        raise ValueError("dest must be [rows, C] with room for the worst case (rows >= T*H*W): N' is only known "
                         "after the kernels have run")
    feat, npatch, tlbr, cnt = quadtree_merge_raw(_video_feature, threshold, temporal_thresh, root_level, weighted_avg,
                                                 head_dim, slow_ver, feat_dest=dest)
    n = cnt[_lib.CNT_OUT]
    return feat[:n], npatch[:n], tlbr[:n]
