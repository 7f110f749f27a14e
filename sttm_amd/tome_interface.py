"""ToMe baseline behind the reference's L1 boundary (`token_merging_utils/tome_interface.py:3-9`).

`get_tome_features(_video_feature, prune_ratio, tome_ver, n_head=1)` keeps the reference's dispatch and its
quirks: "video" works, "frame" raises RuntimeError for T > 1, "snippet" and unknown
versions return None.  The per-video loop (tome_token_merger.py:133-152) runs on the device with no host
synchronisation at all: the number of tokens after every iteration is known on the host in advance.
float32, bfloat16 and float16 inputs (the hook hands over the decoder's hidden states, bf16 in production); 16-bit inputs
follow the reference's per-op rounding to the input dtype (csrc/tome.hip).
"""
import math

import torch

from . import _lib

_DTYPE_CODE = {torch.float32: _lib.STTM_F32, torch.bfloat16: _lib.STTM_BF16, torch.float16: _lib.STTM_F16}


def _tome_video(x_tchw, prune_ratio, n_head):
    if not x_tchw.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; "
                           "there is no CPU fallback")
    if x_tchw.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {x_tchw.dtype} is not supported (float32, bfloat16, float16)")
    dtype = _DTYPE_CODE[x_tchw.dtype]
    lib = _lib.load()
    T, C, H, W = x_tchw.shape
    x = x_tchw.permute(0, 2, 3, 1).reshape(T * H * W, C)           # view for the production layout
    if not x.is_contiguous():
        x = x.contiguous()
    dev = x.device
    n = x.shape[0]
    target = math.ceil(n * (1 - prune_ratio))
    idx = None                                                      # the identity: written by the first step's merge kernel
    size = None
    first = True
    stream = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        while first or n > target:
            first = False
            r = min(n - target, n // 2)
            if r <= 0:
                # the reference's do_nothing(x, mode=None) stand-in is called as merge(x*size, token_idx, mode="sum")
                raise TypeError("do_nothing() got multiple values for argument 'mode' "
                                "(prune_ratio <= 0 is unusable in the reference)")
            nbytes = lib.sttm_tome_workspace_bytes(n, C, int(n_head))
            if nbytes == 0:
                raise ValueError(f"bad ToMe configuration n={n} C={C} n_head={n_head} (or a clip beyond the 2 GiB unit-row limit of include/sttm_hip.h)")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            x_out = torch.empty((n - r, C), dtype=x.dtype, device=dev)
            size_out = torch.empty(n - r, dtype=torch.float32, device=dev)
            idx_out = torch.empty(n - r, dtype=torch.int64, device=dev)
            rc = lib.sttm_tome_step(x.data_ptr(), size.data_ptr() if size is not None else None, idx.data_ptr() if idx is not None else None,
                                    n, C, int(n_head), r, dtype, ws.data_ptr(), nbytes,
                                    x_out.data_ptr(), size_out.data_ptr(), idx_out.data_ptr(), None, None,
                                    stream.cuda_stream)
            _lib.raise_for(rc)
            x, size, idx, n = x_out, size_out, idx_out, n - r
    return x, idx


def get_tome_features(_video_feature, prune_ratio, tome_ver, n_head=1):
    if tome_ver == "frame":
        if _video_feature.shape[0] > 1:
            # tome_per_frame keeps a [1, n, 1] token-index tensor next to [T, n, C] features and torch.gather
            # raises (tome_token_merger.py:52-54)
            raise RuntimeError("Sizes of tensors must match except in dimension 1: tome_ver='frame' is broken in "
                               "the reference for T > 1 (token index batch 1 vs T)")
        return _tome_video(_video_feature, prune_ratio, n_head)
    if tome_ver == "video":
        return _tome_video(_video_feature, prune_ratio, n_head)
    return None            # "snippet" is a stub upstream; unknown versions fall through the reference's if/elif


_side_streams = {}      # device index -> [torch.cuda.Stream, ...]


def get_tome_features_batch(videos, prune_ratio, tome_ver="video", n_head=1, streams=2):
    """Extension (the reference's API is one video per call): ToMe on a LIST of independent videos, results identical to calling
    get_tome_features on each.  A call is a chain of launches per merge iteration -- normalise, the MFMA match, rank, fill, merge -- in which
    everything but the match is a short latency-bound kernel; the videos are dealt out to `streams` side streams (forked from and joined
    back into the current stream with events, nothing synchronises the host) so that the short kernels of one video run in the shadow of
    another video's match.  Returns a list of (features, token_idx)."""
    if not videos:
        return []
    if tome_ver != "video":
        return [get_tome_features(v, prune_ratio, tome_ver, n_head) for v in videos]
    dev = videos[0].device
    if not all(v.is_cuda and v.device == dev for v in videos):
        raise RuntimeError("sttm_amd runs on the GPU only: every video must be a CUDA (ROCm) tensor on one device; there is no CPU fallback")
    ns = max(1, min(int(streams), len(videos)))
    if ns == 1:
        return [_tome_video(v, prune_ratio, n_head) for v in videos]
    with torch.cuda.device(dev):
        pool = _side_streams.setdefault(dev.index, [])
        while len(pool) < ns:
            pool.append(torch.cuda.Stream(device=dev))
        cur = torch.cuda.current_stream(dev)
        fork = torch.cuda.Event()
        fork.record(cur)
        out = [None] * len(videos)
        for k in range(ns):
            pool[k].wait_event(fork)
        for j, v in enumerate(videos):
            st = pool[j % ns]
            with torch.cuda.stream(st):
                x, idx = _tome_video(v, prune_ratio, n_head)
            # the results are handed to the caller's stream: the caching allocator must not recycle them for the side stream first
            x.record_stream(cur)
            idx.record_stream(cur)
            out[j] = (x, idx)
        for k in range(ns):
            done = torch.cuda.Event()
            done.record(pool[k])
            cur.wait_event(done)
    return out
