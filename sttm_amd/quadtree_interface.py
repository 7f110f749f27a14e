"""L1 function boundary of STTM, MI355X edition.

`get_quadtree_features` keeps the signature, argument meaning, return types and error behaviour of the
reference's `token_merging_utils/quadtree_interface.py:5-13` (which forwards to
`quadtree_builder.py:85-235`), but runs entirely in hand-written HIP kernels behind the C ABI of
`libsttm_hip.so`.  PyTorch is used for device memory and the current stream only.
"""
import ctypes
import threading

import torch

from . import _lib

_DTYPE_CODE = {torch.float32: _lib.STTM_F32, torch.bfloat16: _lib.STTM_BF16, torch.float16: _lib.STTM_F16}
_ws_bytes_cache = {}    # (T, H, W, C, dtype, root_level) -> bytes

_seq = [0]
_seq_lock = threading.Lock()


def _next_seq(n=1):
    with _seq_lock:
        first = (_seq[0] % 0x3fff0000) + 1
        _seq[0] = first + n - 1
    return first


_GUARD_TIMEOUT_S = 120.0
# host wait for N': a margin over the device-side bound of the fused label stage's grid barrier (2 s), so that a real barrier
# timeout is READ from the early words (flag bit 31) instead of being pre-empted by the host's own give-up path
_WAIT_TIMEOUT_US = 2_500_000
_NO_FUSE_CALLS = 1000       # calls a stream keeps the two-launch label path after one of its barriers timed out, before re-arming
FLAG_NO_FUSE = 1            # STTM_FLAG_NO_FUSE of include/sttm_hip.h


def _channels_last(x):
    """The production layout is a channels-last VIEW (stride_c == 1); anything else gets one transposing copy (enqueued on the
    current stream, before the kernels that read it)."""
    T, C, H, W = x.shape
    # the spatial kernel addresses one frame with 32-bit byte offsets (csrc/quadtree_spatial.inc): a view whose frame spans
    # 2 GiB or more (exotic strides) is copied too
    frame_span = ((H - 1) * x.stride(2) + (W - 1) * x.stride(3) + C) * x.element_size()
    if x.stride(1) != 1 or (x.data_ptr() % 16) != 0 or frame_span >= 2 ** 31:
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return x


def _workspace_bytes(lib, T, H, W, C, dtype, root_level):
    key = (T, H, W, C, dtype, int(root_level))
    nbytes = _ws_bytes_cache.get(key)
    if nbytes is None:
        nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, dtype, int(root_level))
        if nbytes == 0:
            code = lib.sttm_quadtree_num_levels(H, W, int(root_level))
            _lib.raise_for(code if code < 0 else _lib.ERR_ARG)
        _ws_bytes_cache[key] = nbytes
    return nbytes


def _check_overflow(ovf, cnt):
    if ovf & _lib.OVF_BARRIER_TIMEOUT:
        raise BarrierTimeout("libsttm_hip: the fused label stage's grid barrier timed out (other streams held the CUs): counts=%s" % (cnt,))
    if ovf:
        raise RuntimeError("libsttm_hip: internal list overflow (please report): counts=%s" % (cnt,))


class BarrierTimeout(RuntimeError):
    """The one-launch label stage needs its R column workgroups co-resident; when other work holds the CUs for longer than its
    2 s bound it gives up and says so.  The wrappers below repeat THAT call on the two-launch label path (no residency
    assumption; the `flags` field of the argument block) and keep the calling stream on it for the next _NO_FUSE_CALLS
    calls -- no process-wide switch is touched, other threads and streams are not affected."""


def _with_barrier_retry(st, fn):
    """Run fn(flags) for the stream state `st` (whose lock the caller holds); a grid-barrier timeout repeats it once with the
    two-launch label path."""
    flags = 0
    if st.no_fuse_left > 0:
        st.no_fuse_left -= 1
        flags = FLAG_NO_FUSE
    try:
        return fn(flags)
    except BarrierTimeout as e:
        if flags & FLAG_NO_FUSE:
            raise
        import warnings
        warnings.warn(str(e) + " -- repeating the call on the two-launch label path (this stream keeps it for the next "
                      "%d calls)" % _NO_FUSE_CALLS)
        st.no_fuse_left = _NO_FUSE_CALLS
        return fn(FLAG_NO_FUSE)


class _StreamState:
    """Everything one (device, stream) needs call after call, for the one-video AND the batch entry point: the argument block
    of sttm_quadtree_merge_packed (only the fields that change are rewritten), the pinned landing pads of N' (classic counts +
    the early per-column words; rows for the videos of a batch), scratch, and the ONE lock that serialises host threads which
    share the stream."""
    __slots__ = ("args", "args_ptr", "pinned", "host_ptr", "early_ptr", "host_view", "out2", "out2_ptr", "ws", "counts", "key", "lock",
                 "idx", "handle", "evicted", "no_fuse_left", "batch_pinned", "batch_early", "retired", "n_early_out")

    def __init__(self, idx, handle):
        # one pinned block: int32[8] classic counts, then uint64[EARLY_SLOTS] early words (8-byte aligned at byte 32)
        self.pinned = torch.zeros(4 + _lib.EARLY_SLOTS, dtype=torch.int64).pin_memory()
        self.host_ptr = self.pinned.data_ptr()
        self.early_ptr = self.host_ptr + 32
        self.host_view = self.pinned[:4].view(torch.int32)
        self.args = _lib.MergeArgs()
        self.args_ptr = ctypes.addressof(self.args)
        self.args.counts_host = self.host_ptr
        self.args.early_host = self.early_ptr
        self.args.stride_c = 1
        self.out2 = (ctypes.c_int32 * 2)()
        self.out2_ptr = ctypes.addressof(self.out2)
        self.ws = None
        self.counts = None
        self.key = None
        self.lock = threading.Lock()
        self.idx, self.handle = idx, handle
        self.evicted = False
        self.no_fuse_left = 0
        self.batch_pinned = None        # int32 [rows, CNT_SLOTS] pinned: classic counts of the videos of a batch call
        self.batch_early = None         # int64 [rows, EARLY_SLOTS] pinned: the early per-column words of the videos of a batch call
        self.n_early_out = ctypes.c_int(0)
        self.retired = []               # outgrown pinned pads: kept until this state itself is retired (the stream may still write them)

    def reserve(self, dev, nbytes, rows):
        """Scratch for `nbytes` and device counts for `rows` videos (grown, never shrunk; allocated on this state's stream, so the
        caching allocator orders the release of an outgrown block behind the work queued on it)."""
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.key = None
        if self.counts is None or self.counts.shape[0] < rows:
            self.counts = torch.empty((max(rows, 16), _lib.CNT_SLOTS), dtype=torch.int32, device=dev)
            self.key = None

    def batch_rows(self, rows):
        if self.batch_pinned is None or self.batch_pinned.shape[0] < rows:
            if self.batch_pinned is not None:
                self.retired.append(self.batch_pinned)
                self.retired.append(self.batch_early)
            self.batch_pinned = torch.zeros((max(rows, 32), _lib.CNT_SLOTS), dtype=torch.int32).pin_memory()
            self.batch_early = torch.zeros((max(rows, 32), _lib.EARLY_SLOTS), dtype=torch.int64).pin_memory()
        return self.batch_pinned, self.batch_early

    def stream_idle(self):
        """True when everything queued on this state's stream has completed (its pinned pads are then no longer written).
        The stream is re-wrapped from its raw handle: torch's own streams come from a fixed per-device pool and are never destroyed, so
        the handle of a torch stream stays valid for the life of the process; a foreign stream (torch.cuda.ExternalStream of another
        runtime user) that its owner has destroyed is the owner's contract to keep alive until the states created on it are retired --
        `sttm_amd.quadtree_interface.drop_stream_state(stream)` retires one explicitly."""
        try:
            s = torch.cuda.default_stream(self.idx) if not self.handle else torch.cuda.ExternalStream(self.handle, device=self.idx)
            return bool(s.query())
        except Exception:      # noqa: BLE001 -- the stream was destroyed: nothing of it is in flight
            return True

    def drain(self):
        try:
            s = torch.cuda.default_stream(self.idx) if not self.handle else torch.cuda.ExternalStream(self.handle, device=self.idx)
            s.synchronize()
        except Exception:      # noqa: BLE001
            pass


class _StateCache:
    """(device index, raw stream handle) -> _StreamState, least-recently-USED first out (a hit refreshes the entry, so the
    default stream of a long-lived server is not the first to go).  A state whose lock is held -- a thread is inside a call on
    it -- is never evicted.  An evicted state goes to a graveyard that keeps its pinned pads alive until ITS stream has drained
    (the group-mean kernel of its last call writes the classic counts after the host has already returned on the early words);
    nothing blocks and no device-wide synchronisation happens on this path.  Pinning memory and querying streams happen outside
    the table's lock."""

    def __init__(self, limit=8):
        import collections
        import os
        self.limit = int(os.environ.get("STTM_WS_CACHE", limit))
        self._d = collections.OrderedDict()
        self._mu = threading.Lock()
        self._graveyard = []

    def __len__(self):
        return len(self._d)

    def __contains__(self, key):
        return key in self._d

    def graveyard_size(self):
        return len(self._graveyard)

    def get_or_create(self, idx, handle):
        key = (idx, handle)
        with self._mu:
            st = self._d.get(key)
            if st is not None:
                self._d.move_to_end(key)
                return st
        fresh = _StreamState(idx, handle)            # pins host memory: outside the lock
        evicted = []
        with self._mu:
            st = self._d.get(key)
            if st is not None:                       # another thread was faster
                self._d.move_to_end(key)
                return st
            self._d[key] = fresh
            if len(self._d) > self.limit:
                for k in list(self._d):
                    if len(self._d) <= self.limit:
                        break
                    cand = self._d[k]
                    if cand is fresh or not cand.lock.acquire(False):
                        continue                     # in use right now: the next least-recently-used one goes instead
                    cand.evicted = True              # a thread that already holds a reference re-resolves its state
                    del self._d[k]
                    cand.lock.release()
                    evicted.append(cand)
        if evicted:
            self._retire(evicted)
        return fresh

    def _retire(self, states):
        with self._mu:
            self._graveyard.extend(states)
            pending = list(self._graveyard)
        done = [g for g in pending if g.stream_idle()]
        with self._mu:
            self._graveyard = [g for g in self._graveyard if g not in done]
            overflow = self._graveyard[:-32] if len(self._graveyard) > 32 else []
        for g in overflow:                            # a caller that burns through streams faster than they drain (rare)
            g.drain()
        if overflow:
            with self._mu:
                self._graveyard = [g for g in self._graveyard if g not in overflow]


_states = _StateCache(8)          # (device index, raw stream handle) -> _StreamState


def drop_stream_state(stream):
    """Retire the per-stream state of `stream` (a torch.cuda.Stream / ExternalStream) NOW: waits for the work queued on it, then frees
    its scratch and pinned landing pads.  For owners of foreign streams, to be called before they destroy the stream."""
    key = (stream.device.index, stream.cuda_stream)
    with _states._mu:
        st = _states._d.pop(key, None)
    if st is not None:
        with st.lock:
            st.evicted = True
        stream.synchronize()
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _state_for(dev, idx, handle):
    return _states.get_or_create(idx, handle)


def _acquire_state(dev):
    """The state of (current device, current stream) with its lock HELD.  Two host threads that issue merges on the SAME stream
    (the default stream of a threaded inference server, say) share that stream's scratch and landing pads: their calls are
    serialised here.  Threads on their own streams run concurrently."""
    idx = torch.cuda.current_device()
    handle = _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
    while True:
        st = _states.get_or_create(idx, handle)
        lock = st.lock
        if not lock.acquire(False) and not lock.acquire(timeout=_GUARD_TIMEOUT_S):
            raise RuntimeError("sttm_amd: waited %.0f s for another host thread inside a merge call on this same stream "
                               "(give every thread its own torch.cuda.Stream: the scratch and the pinned counts are per stream)"
                               % _GUARD_TIMEOUT_S)
        if not st.evicted:
            return st
        lock.release()              # evicted between the lookup and the lock: resolve again


def quadtree_merge_raw(x, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver=False, return_ctx=False,
                       feat_dest=None, events=None, side_tensors=None, side_sum_mode=False, want_idx=False):
    """Launch the merge of one video and return the worst-case-sized outputs plus the host counts (only the slots CNT_OUT and
    CNT_OVERFLOW are filled in on the host; the diagnostic counters stay in the device `counts` tensor of the returned context
    and are complete once the stream has drained).
    x: logical [T, C, H, W] CUDA tensor (any strides; channels-last views run zero-copy).
    events: optional _lib.KernelEvents -- the library records them around its kernels (per-kernel timing).
    side_tensors: optional list of [T, Cv, H, W] tensors (RoPE cos / sin of the position-embedding ablation) pooled over the
    nodes and groups of THIS merge with sttm_quadtree_apply -- inside the same critical section: the apply reads "the merge that
    ran last on this stream with this workspace", so no other host thread on the stream may get in between.  Their pooled
    [T*H*W, Cv] outputs (rows [0, N') valid) are appended to the result as a list.
    want_idx: also return the merged tokens' 1-d indices t*H*W + y1*W + x1 (int32 [T*H*W], rows [0, N') valid), written by the
    group-mean kernel -- appended last.

    Host path (round 3): the device chain of one call is ~80 us and the next call cannot be issued before this one knows N', so
    every microsecond between "N' arrived" and "next spatial kernel submitted" is device idle time.  Hence: no device context
    switch when the tensor's device is already current, the raw stream handle instead of a Stream object, one argument block
    per (device, stream) of which only the changing fields are rewritten, N' read from the early per-column words."""
    if not x.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; "
                           "there is no CPU fallback")
    if x.dim() != 4:
        raise ValueError("expected a [T, C, H, W] tensor")
    dtype = _DTYPE_CODE.get(x.dtype)
    if dtype is None:
        raise NotImplementedError(f"dtype {x.dtype} is not supported (float32, bfloat16, float16)")
    dev = x.device
    idx = dev.index
    if idx is None or torch.cuda.current_device() != idx:
        with torch.cuda.device(dev):
            return _merge_on_current_device(x, dtype, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver,
                                            return_ctx, feat_dest, events, side_tensors, side_sum_mode, want_idx)
    return _merge_on_current_device(x, dtype, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver,
                                    return_ctx, feat_dest, events, side_tensors, side_sum_mode, want_idx)


def _merge_on_current_device(x, dtype, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver, return_ctx,
                             feat_dest, events, side_tensors, side_sum_mode, want_idx=False):
    st = _acquire_state(x.device)
    try:
        out = _with_barrier_retry(st, lambda flags: _merge_locked(st, x, dtype, threshold, temporal_thresh, root_level, weighted_avg,
                                                                   head_dim, slow_ver, feat_dest, events, flags, want_idx))
        feat, npatch, tlbr, cnt, xc, idx1d = out
        ctx = (xc, st.ws, st.counts, dtype, _StreamHandle(st.handle, x.device))
        sides = None
        if side_tensors is not None:
            sides = [_apply_side_tensor(v, ctx, root_level, side_sum_mode) for v in side_tensors]
    finally:
        st.lock.release()
    res = (feat, npatch, tlbr, cnt)
    if return_ctx:
        res += (ctx,)
    if sides is not None:
        res += (sides,)
    if want_idx:
        res += (idx1d,)
    return res


def _merge_locked(st, x, dtype, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver, feat_dest, events, flags,
                  want_idx=False):
    """One merge on the stream state `st`, whose lock the caller holds."""
    lib = _lib.load()
    dev = x.device
    T, C, H, W = x.shape
    sT, sC, sH, sW = x.stride()
    ptr = x.data_ptr()
    # the production layout is a channels-last VIEW (stride_c == 1); anything else -- or a frame spanning >= 2 GiB, which
    # the spatial kernel's 32-bit frame offsets cannot address -- gets one transposing copy on the current stream
    if sC != 1 or (ptr & 15) or ((H - 1) * sH + (W - 1) * sW + C) * x.element_size() >= 2 ** 31:
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        sT, sC, sH, sW = x.stride()
        ptr = x.data_ptr()
    head = 0 if head_dim is None else int(head_dim)
    N = T * H * W
    a = st.args
    key = (T, C, H, W, dtype, sT, sH, sW, threshold, temporal_thresh, root_level, weighted_avg, head, slow_ver)
    if key != st.key:
        nbytes = _workspace_bytes(lib, T, H, W, C, dtype, root_level)
        st.reserve(dev, nbytes, 16)
        a.stride_t, a.stride_h, a.stride_w = sT, sH, sW
        a.T, a.C, a.H, a.W, a.dtype = T, C, H, W, dtype
        a.threshold, a.temporal_thresh = float(threshold), float(temporal_thresh)
        a.root_level, a.weighted_avg, a.head_dim, a.slow_ver = int(root_level), int(bool(weighted_avg)), head, int(bool(slow_ver))
        a.workspace, a.workspace_bytes = st.ws.data_ptr(), st.ws.numel()
        a.counts = st.counts.data_ptr()
        a.stream = st.handle
        st.key = key
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
    idx1d = torch.empty(N, dtype=torch.int32, device=dev) if want_idx else None
    a.idx_out = idx1d.data_ptr() if want_idx else None
    seq = _next_seq()
    a.x, a.feat_out, a.npatch_out, a.tlbr_out, a.seq = ptr, feat.data_ptr(), npatch.data_ptr(), tlbr.data_ptr(), seq
    a.events = events.pointer() if events is not None else None
    a.flags = flags
    rc = lib.sttm_quadtree_merge_packed(st.args_ptr)
    if rc < 0:
        st.key = None
        _lib.raise_for(rc)
    # Output sizes are data dependent, so the host must learn N' -- but only N': every column of the label stage reports its
    # survivors into pinned memory as its last action (fallback: the group-mean kernel's first workgroup publishes the counts)
    # and we wait on that, returning while the feature gather is still running.
    if lib.sttm_wait_counts_early(st.host_ptr, st.early_ptr, a.n_early, seq, _WAIT_TIMEOUT_US, st.out2_ptr) != 0:
        stream = torch.cuda.current_stream(dev)
        st.host_view.copy_(st.counts[0], non_blocking=True)  # fallback: classic D2H + stream sync
        stream.synchronize()
        h = st.host_view.tolist()
        st.out2[0], st.out2[1] = h[_lib.CNT_OUT], h[_lib.CNT_OVERFLOW]
    n_out, ovf = st.out2[0], st.out2[1]
    cnt = [0] * _lib.CNT_SLOTS
    cnt[_lib.CNT_OUT], cnt[_lib.CNT_OVERFLOW] = n_out, ovf
    if ovf:
        _check_overflow(ovf, cnt)
    return feat, npatch, tlbr, cnt, x, idx1d


class _StreamHandle:
    """What the context tuple of quadtree_merge_raw carries as its stream: the raw handle (all the C ABI needs)."""
    __slots__ = ("cuda_stream", "device")

    def __init__(self, handle, device):
        self.cuda_stream, self.device = handle, device


def _apply_side_tensor(v, ctx, root_level, sum_mode):
    """Pool a side tensor (RoPE cos or sin, logical [T, Cv, H, W]) over the nodes / groups of the merge that just ran."""
    x, ws, counts, dtype_feat, stream = ctx
    lib = _lib.load()
    if v.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {v.dtype} is not supported")
    T, Cv, H, W = v.shape
    v = _channels_last(v)
    out = torch.empty((T * H * W, Cv), dtype=v.dtype, device=v.device)
    rc = lib.sttm_quadtree_apply(v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), v.stride(3), T, Cv, H, W,
                                 _DTYPE_CODE[v.dtype], int(bool(sum_mode)), x.shape[1], dtype_feat, int(root_level),
                                 ws.data_ptr(), ws.numel(), counts.data_ptr(), out.data_ptr(), stream.cuda_stream)
    _lib.raise_for(rc)
    return out


def get_quadtree_features_batch(videos, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                                slow_ver=False, head_dim=None, events=None, out=None):
    """Extension (not in the reference, whose API is one video per call): merge a LIST of videos and return the list of
    (features, num_patches, tlbr) triples -- results identical to calling get_quadtree_features on each.

    Videos of one shape / dtype / stride set go through sttm_quadtree_merge_batch together.  Default (stage-skewed form, round 5): the
    library deals them out to `batch_streams` INTERNAL streams (3; created on first use per host thread and device, released by
    `release_streams()`) in launch sets of `batch_sub` videos (8), forked from and joined back into the current stream, so that
    consecutive launch sets sit in different stages at any moment -- the latency-bound label stage and the launch ramps of one set run
    under the bandwidth-bound kernels of the others.  `_lib.configure(batch_streams=0)` selects the lockstep form (every kernel once per
    group of videos, on the caller's stream only).  The host waits for the per-video token counts only after everything has been
    enqueued.  Same per-stream state (lock, scratch, landing pads) as the one-video call.

    Memory: N' is known only after the kernels ran, so every video gets worst-case output rows ([T*H*W, C] ...); without `out` a call
    allocates them (96 headline videos: 10 GB for 4.4 GB of results, alive as long as the returned views).  `out = (features [n, T*H*W, C],
    num_patches [n, T*H*W] int32, tlbr [n, T*H*W, 5] int32)` -- CALLER-OWNED blocks for n >= len(videos) videos of ONE shape / dtype,
    reused call after call -- makes the call allocation-free; the results are leading views into them."""
    if not videos:
        return []
    dev = videos[0].device
    if not all(v.is_cuda and v.device == dev for v in videos):
        raise RuntimeError("sttm_amd runs on the GPU only: every video must be a CUDA (ROCm) tensor on one device; "
                           "there is no CPU fallback")
    with torch.cuda.device(dev):
        st = _acquire_state(dev)
        try:
            return _with_barrier_retry(st, lambda flags: _batch_locked(st, videos, dev, threshold, temporal_thresh, root_level,
                                                                        weighted_avg, slow_ver, head_dim, events, flags, out))
        finally:
            st.lock.release()


def release_streams():
    """Destroy the internal streams / events the batch entry point created for the calling host thread (C ABI: sttm_release_streams)."""
    return _lib.load().sttm_release_streams()


def _batch_locked(st, videos, dev, threshold, temporal_thresh, root_level, weighted_avg, slow_ver, head_dim, events, flags, dest=None):
    lib = _lib.load()
    head = 0 if head_dim is None else int(head_dim)
    out = [None] * len(videos)
    groups = {}
    vids = []
    for j, x in enumerate(videos):
        if x.dim() != 4 or x.dtype not in _DTYPE_CODE:
            raise ValueError("expected [T, C, H, W] float32 / bfloat16 / float16 tensors")
        x = _channels_last(x)
        vids.append(x)
        groups.setdefault((tuple(x.shape), x.dtype, tuple(x.stride())), []).append(j)
    pend = []
    rows_used = 0
    total = len(videos)
    # one workspace block per video of the call (all groups): sized for the largest
    per_video = 0
    for (shape, dt, _), ids in groups.items():
        T, C, H, W = shape
        per_video = max(per_video, _workspace_bytes(lib, T, H, W, C, _DTYPE_CODE[dt], root_level))
    per_video = (per_video + 255) // 256 * 256
    st.reserve(dev, per_video * total, total)
    host, early = st.batch_rows(total)
    host_base, early_base = host.data_ptr(), early.data_ptr()
    ws, counts = st.ws, st.counts
    for (shape, dt, strides), ids in groups.items():
        T, C, H, W = shape
        N = T * H * W
        n = len(ids)
        # the outputs of a group in THREE allocations (worst-case rows per video, returned as leading views like the one-video call):
        # one torch.empty per tensor and video kept the host busy for ~7 us per video between two calls, with the device idle
        if dest is not None:
            if len(groups) != 1:
                raise ValueError("out= takes videos of ONE shape / dtype / stride set")
            feats, npatches, tlbrs = dest
            ok = (feats.is_cuda and feats.device == dev and feats.dtype == dt and feats.is_contiguous() and tuple(feats.shape[1:]) == (N, C)
                  and feats.shape[0] >= n and npatches.dtype == torch.int32 and npatches.is_contiguous() and tuple(npatches.shape) == (feats.shape[0], N)
                  and npatches.device == dev and tlbrs.dtype == torch.int32 and tlbrs.is_contiguous() and tlbrs.device == dev
                  and tuple(tlbrs.shape) == (feats.shape[0], N, 5))
            if not ok:
                raise ValueError(f"out= must be contiguous (features [n, {N}, {C}] {dt}, num_patches [n, {N}] int32, tlbr [n, {N}, 5] int32) on "
                                 f"{dev} with n >= {n}")
        else:
            feats = torch.empty((n, N, C), dtype=dt, device=dev)
            npatches = torch.empty((n, N), dtype=torch.int32, device=dev)
            tlbrs = torch.empty((n, N, 5), dtype=torch.int32, device=dev)
        eb = feats.element_size()
        f0, p0, t0 = feats.data_ptr(), npatches.data_ptr(), tlbrs.data_ptr()
        vp = ctypes.c_void_p * n
        seq = _next_seq(n)
        rc = lib.sttm_quadtree_merge_batch(
            n, vp(*[vids[j].data_ptr() for j in ids]), strides[0], strides[1], strides[2], strides[3], T, C, H, W, _DTYPE_CODE[dt],
            float(threshold), float(temporal_thresh), int(root_level), int(bool(weighted_avg)), head, int(bool(slow_ver)),
            ws.data_ptr() + rows_used * per_video, per_video,
            vp(*range(f0, f0 + n * N * C * eb, N * C * eb)), vp(*range(p0, p0 + n * N * 4, N * 4)), vp(*range(t0, t0 + n * N * 20, N * 20)),
            counts[rows_used].data_ptr(), host_base + rows_used * 4 * _lib.CNT_SLOTS, seq,
            events.pointer() if events is not None else None, st.handle, flags,
            early_base + rows_used * 8 * _lib.EARLY_SLOTS, ctypes.byref(st.n_early_out))
        _lib.raise_for(rc)
        n_early = st.n_early_out.value
        for k, j in enumerate(ids):
            pend.append((j, feats[k], npatches[k], tlbrs[k], seq + k, rows_used + k, n_early))
        rows_used += n
    out2, out2_ptr = st.out2, st.out2_ptr
    for j, feat, npatch, tlbr, seq, row, n_early in pend:
        # N' of every video from pinned memory: the label stage's columns report it (one kernel before the feature gather ends), the
        # group-mean kernel's first workgroup is the fallback; no stream synchronisation
        if lib.sttm_wait_counts_early(host_base + row * 4 * _lib.CNT_SLOTS, early_base + row * 8 * _lib.EARLY_SLOTS, n_early, seq,
                                      _WAIT_TIMEOUT_US, out2_ptr) != 0:
            host[row].copy_(counts[row], non_blocking=True)  # fallback: classic D2H + stream sync
            torch.cuda.current_stream(dev).synchronize()
            cnt = host[row].tolist()
            out2[0], out2[1] = cnt[_lib.CNT_OUT], cnt[_lib.CNT_OVERFLOW]
        n_out, ovf = out2[0], out2[1]
        if ovf:
            cnt = [0] * _lib.CNT_SLOTS
            cnt[_lib.CNT_OUT], cnt[_lib.CNT_OVERFLOW] = n_out, ovf
            _check_overflow(ovf, cnt)
        out[j] = _sized(feat, npatch, tlbr, n_out, owned=dest is None)
    return out


_exact_outputs = [bool(int(__import__("os").environ.get("STTM_EXACT_OUTPUTS", "0")))]


def set_exact_outputs(flag):
    """Opt-in compaction.  N' is only known after the kernels have run, so the outputs are allocated for the worst case
    ([T*H*W, C]: 103 MB for 46 MB used at the 128-frame headline, 578 MB at T=180 C=8192 bf16) and returned as leading VIEWS --
    the whole block stays alive as long as the caller holds the result.  With exact outputs the three results are copied into
    exact-size tensors (one extra read + write of N' rows, stream-ordered) and the worst-case block goes back to the caching
    allocator at once, like the reference's exact-size returns (quadtree_builder.py:198-226).  The copy-free alternative is
    get_quadtree_features_into (the caller owns the destination).  Also settable with STTM_EXACT_OUTPUTS=1."""
    _exact_outputs[0] = bool(flag)


def _sized(feat, npatch, tlbr, n, owned=True):
    if _exact_outputs[0] and owned:
        return feat[:n].clone(), npatch[:n].clone(), tlbr[:n].clone()
    return feat[:n], npatch[:n], tlbr[:n]


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
        return _sized(feat, npatch, tlbr, cnt[_lib.CNT_OUT])
    # position-embedding ablation (quadtree_attn_monkey_patch_for_abl_pos.py): cos / sin ride along the same tree
    cos, sin = pos_embs
    if temporal_thresh <= 0 and not pos_emb_weighted_avg:
        # quirk Q11 of the reference: `pos_embs_cos` is only assigned on the temporal or the weighted path
        raise UnboundLocalError("local variable 'pos_embs_cos' referenced before assignment (reference behaviour for "
                                "pos_embs with temporal_thresh <= 0 and pos_emb_weighted_avg=False)")
    # the merge and the two poolings that read its node / group tables run under ONE hold of the stream's lock
    feat, npatch, tlbr, cnt, (out_cos, out_sin) = quadtree_merge_raw(
        _video_feature, threshold, temporal_thresh, root_level, weighted_avg, head_dim, slow_ver,
        side_tensors=[cos, sin], side_sum_mode=pos_emb_weighted_avg)
    n = cnt[_lib.CNT_OUT]
    if _exact_outputs[0]:
        out_cos, out_sin = out_cos[:n].clone(), out_sin[:n].clone()
    else:
        out_cos, out_sin = out_cos[:n], out_sin[:n]
    return _sized(feat, npatch, tlbr, n) + ((out_cos, out_sin),)


def get_quadtree_features_into(dest, _video_feature, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                               slow_ver=False, head_dim=None, return_idx=False):
    """get_quadtree_features with a caller-owned destination: the merged features are written to dest[:N'] directly
    (dest: contiguous [rows >= T*H*W, C] tensor, e.g. a row window of the NEW hidden-state buffer), so the caller's
    `torch.cat([system, merged, instruction])` (quadtree_attn_monkey_patch.py:105) needs no copy of the merged rows.
    Nothing of `dest` beyond row N' is written.  dest must not overlap the input.
    Returns (dest[:N'], num_patches [N'], tlbr [N', 5]); with return_idx also the merged tokens' 1-d indices t*H*W + y1*W + x1
    (int32 [N'], what the hook computes from tlbr at quadtree_attn_monkey_patch.py:103-104), written by the kernels."""
    T, C, H, W = _video_feature.shape
    if dest.dim() != 2 or dest.size(0) < T * H * W:
        raise ValueError("dest must be [rows, C] with room for the worst case (rows >= T*H*W): N' is only known "
                         "after the kernels have run")
    out = quadtree_merge_raw(_video_feature, threshold, temporal_thresh, root_level, weighted_avg,
                             head_dim, slow_ver, feat_dest=dest, want_idx=return_idx)
    feat, npatch, tlbr, cnt = out[:4]
    n = cnt[_lib.CNT_OUT]
    if _exact_outputs[0]:
        res = (feat[:n], npatch[:n].clone(), tlbr[:n].clone())        # (the features already live in the caller's buffer)
        return res + ((out[4][:n].clone(),) if return_idx else ())
    return (feat[:n], npatch[:n], tlbr[:n]) + ((out[4][:n],) if return_idx else ())


get_quadtree_features_into.returns_idx = True        # patch_hooks._merge_concat asks for merged_token_1d_idx along with the merge


_POOL_MODES = {"average": 0, "max": 1, "bilinear": 2}


def get_quadtree_features_from_pooled_input(image_feature, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                                            slow_ver=False, head_dim=None, *, stride=2, mode="bilinear", width=-1,
                                            num_patches_per_side=None, force_fused=False):
    """get_quadtree_features on the tokens BEFORE `get_2dPool`: `image_feature` is the projected vision-token map
    [T, side*side, C] (LLaVA-Video: 27 x 27 = 729 tokens per frame, llava/eval/video_feat_llavavideo.py:89-95 after the projector) and
    the call returns what the reference computes in two steps,

        pooled = get_2dPool(image_feature, stride)                         # llava/model/llava_arch.py:173-198 (mode = mm_spatial_pool_mode)
        get_quadtree_features(rearrange(pooled, "t (h w) c -> t c h w"), threshold, temporal_thresh, root_level, ...)

    with the pooling FUSED into the spatial kernel's leaf load (`sttm_quadtree_merge_pooled`): the 4 source tokens of every leaf are read
    once, the pooled map is never written or re-read.  Shapes the fused kernel does not cover (trees that are not 3 levels deep, per-head
    cosine, odd channel counts, strides other than 2 for average / max) run the same two steps on the device: `sttm_pool2d`, then the merge.
    Returns (features [N', C], num_patches [N'], tlbr [N', 5]) over the pooled grid."""
    import math
    x = image_feature
    if stride == 1:                                                         # get_2dPool's identity case (:174-175)
        T, n_tok, C = x.shape
        side = num_patches_per_side if (width == -1 and num_patches_per_side is not None) else (int(round(math.sqrt(n_tok))) if width == -1 else width)
        return get_quadtree_features(x.reshape(T, side, n_tok // side, C).permute(0, 3, 1, 2), threshold, temporal_thresh, root_level,
                                     weighted_avg, False, slow_ver, head_dim)
    if mode not in _POOL_MODES:
        raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")        # :194-195
    if not x.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; there is no CPU fallback")
    if x.dim() != 3:
        raise ValueError("expected a [num_frames, num_tokens, C] tensor")
    dtype = _DTYPE_CODE.get(x.dtype)
    if dtype is None:
        raise NotImplementedError(f"dtype {x.dtype} is not supported (float32, bfloat16, float16)")
    T, n_tok, C = x.shape
    if width == -1:
        side_h = side_w = num_patches_per_side if num_patches_per_side is not None else int(round(math.sqrt(n_tok)))
    else:
        side_h = side_w = width
    if side_h * side_w != n_tok:                                            # the reference's .view() fails the same way (:181)
        raise RuntimeError("shape '[%d, %d, %d, -1]' is invalid for input of size %d" % (T, side_h, side_w, x.numel()))
    lib = _lib.load()
    m = _POOL_MODES[mode]
    H, W = lib.sttm_pool2d_out_side(side_h, int(stride), m), lib.sttm_pool2d_out_side(side_w, int(stride), m)
    if H < 1 or W < 1:
        _lib.raise_for(_lib.ERR_ARG)
    x = x.contiguous()
    eb = x.element_size()
    # rows below 4 KB (two waves per root cell) do not keep enough source rows in flight: measured bf16 C = 1024 112 us fused against
    # 105 us in two steps, C = 2048 158 / 161, fp32 C = 1024 134 / 157, bf16 C = 3584 253 / 263 (tools/bench_pool_fused.py); force=True
    # in the tests runs the fused kernel on every shape it supports
    fused = (head_dim is None and (mode == "bilinear" or int(stride) == 2) and (C * eb) % 16 == 0 and x.data_ptr() % 16 == 0
             and max(H, W) <= 64 and lib.sttm_quadtree_num_levels(H, W, int(root_level)) == 3 and (force_fused or C * eb >= 4096))
    if not fused:
        from .upstream import get_2dPool
        pooled = get_2dPool(x, stride=stride, width=side_w, mode=mode)
        return get_quadtree_features(pooled.reshape(T, H, W, C).permute(0, 3, 1, 2), threshold, temporal_thresh, root_level,
                                     weighted_avg, False, slow_ver, head_dim)
    dev = x.device
    N = T * H * W
    def run(flags):
        nbytes = _workspace_bytes(lib, T, H, W, C, dtype, root_level)
        st.reserve(dev, nbytes, 16)
        st.key = None                                                   # (the argument block of the plain merge is not what ran last)
        feat = torch.empty((N, C), dtype=x.dtype, device=dev)
        npatch = torch.empty(N, dtype=torch.int32, device=dev)
        tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev)
        seq = _next_seq()
        rc = lib.sttm_quadtree_merge_pooled(x.data_ptr(), T, side_h, side_w, C, dtype, m, int(stride), float(threshold), float(temporal_thresh),
                                            int(root_level), int(bool(weighted_avg)), int(bool(slow_ver)), st.ws.data_ptr(), st.ws.numel(),
                                            feat.data_ptr(), npatch.data_ptr(), tlbr.data_ptr(), st.counts.data_ptr(), st.host_ptr, seq, st.handle,
                                            flags)
        _lib.raise_for(rc)
        if lib.sttm_wait_counts(st.host_ptr, seq, _WAIT_TIMEOUT_US) != 0:
            st.host_view.copy_(st.counts[0], non_blocking=True)        # fallback: classic D2H + stream sync
            torch.cuda.current_stream(dev).synchronize()
        cnt = st.host_view.tolist()
        _check_overflow(cnt[_lib.CNT_OVERFLOW], cnt)                    # (a grid-barrier timeout: _with_barrier_retry repeats the call)
        return feat, npatch, tlbr, cnt

    with torch.cuda.device(dev):
        st = _acquire_state(dev)
        try:
            feat, npatch, tlbr, cnt = _with_barrier_retry(st, run)
        finally:
            st.lock.release()
    return _sized(feat, npatch, tlbr, cnt[_lib.CNT_OUT])


def temporal_merge_nodes(node_features, node_tlbr, temporal_thresh, weighted_avg=False, head_dim=None, *, grid, root_level, slow_ver=False):
    """Extension: the temporal stage alone on a node list in ANY order (C ABI: `sttm_temporal_merge`).

    node_features [N, C] (float32 / bfloat16 / float16) and node_tlbr [N, 5] = (t, y1, x1, y2, x2) are nodes of the quadtree partition
    `grid = (T, H, W)` / `root_level` (what `get_quadtree_features(..., temporal_thresh=-1)` emits; every box must be a cell of that
    partition and the boxes of a frame must be disjoint -- anything else raises).  Returns (features [N', C], num_patches [N'] = box
    areas summed over each group, tlbr [N', 5]) ordered by (t, y1, x1); a group is represented by its first node in that order.  The
    pair filter keeps `sim >= temporal_thresh` for any sign of the threshold, like quadtree_temporal_merger.py:70-71."""
    x, tl = node_features, node_tlbr
    if not x.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; there is no CPU fallback")
    if x.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {x.dtype} is not supported (float32, bfloat16, float16)")
    T, H, W = (int(v) for v in grid)
    N, C = x.shape
    if tl.shape != (N, 5):
        raise ValueError(f"tlbr must be [N, 5]; got {tuple(tl.shape)} for N = {N}")
    head = 0 if head_dim is None else int(head_dim)
    lib = _lib.load()
    dev = x.device
    x = x.contiguous()
    tl = tl.to(device=dev, dtype=torch.int32).contiguous()
    code = _DTYPE_CODE[x.dtype]
    with torch.cuda.device(dev):
        nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, code, int(root_level))
        if nbytes == 0:
            _lib.raise_for(_lib.ERR_INDEX if "root_level" in _lib.last_error() else _lib.ERR_ARG)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rows = max(N, 1)
        feat = torch.empty((rows, C), dtype=x.dtype, device=dev)
        npatch = torch.empty(rows, dtype=torch.int32, device=dev)
        tlbr = torch.empty((rows, 5), dtype=torch.int32, device=dev)
        counts = torch.zeros(_lib.CNT_SLOTS, dtype=torch.int32, device=dev)
        rc = lib.sttm_temporal_merge(x.data_ptr(), tl.data_ptr(), N, T, C, H, W, code, float(temporal_thresh), int(root_level),
                                     1 if weighted_avg else 0, head, 1 if slow_ver else 0, ws.data_ptr(), nbytes,
                                     feat.data_ptr(), npatch.data_ptr(), tlbr.data_ptr(), counts.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream)
        _lib.raise_for(rc)
        cnt = counts.cpu().tolist()
    if cnt[_lib.CNT_OVERFLOW]:
        raise RuntimeError(f"sttm_temporal_merge: invalid node list (overflow flags {cnt[_lib.CNT_OVERFLOW]}): a box outside the "
                           f"{H} x {W} grid, a box that is not a cell of the root_level = {root_level} partition, a duplicated origin, "
                           "overlapping boxes, or more nodes than leaves in a root cell")
    n = cnt[_lib.CNT_OUT]
    return feat[:n], npatch[:n], tlbr[:n]


def _merging_as_the_reference(x, tl, temporal_thresh, num_patches, weighted_avg, head_dim, cos, sin, grid, root_level, slow_ver):
    if cos is not None or sin is not None:
        raise NotImplementedError("position embeddings ride along the fused merge only (get_quadtree_features(..., pos_embs=...))")
    if grid is None or root_level is None:
        raise TypeError("sttm_amd needs grid=(T, H, W) and root_level=... (keyword arguments): the nodes go into the per-root-cell "
                        "tables of the fused merge, which the reference function -- comparing every pair of boxes -- does not need")
    tl = tl.to(device=x.device)
    # the reference's representative is the LOWEST INDEX of a group and its output keeps the input order: that equals this
    # implementation's (t, y1, x1) order exactly when the list is sorted that way, as quadtree_build_video passes it
    # (quadtree_builder.py:198-203)
    if tl.shape[0] > 1:
        T, H, W = (int(v) for v in grid)
        key = (tl[:, 0].long() * H + tl[:, 1].long()) * W + tl[:, 2].long()
        if not bool((key[1:] > key[:-1]).all()):
            raise ValueError("the node list must be sorted by (t, y1, x1) without duplicates, as quadtree_build_video passes it "
                             "(temporal_merge_nodes takes any order and represents a group by its first node in that order)")
    if num_patches is not None:
        area = (tl[:, 3] - tl[:, 1]) * (tl[:, 4] - tl[:, 2])
        if not torch.equal(num_patches.to(device=x.device, dtype=area.dtype), area):
            raise NotImplementedError("quadtree_num_patches_per_node must equal the box areas (what quadtree_build_video passes, "
                                      "quadtree_builder.py:206-209): the kernels derive the patch counts from the boxes")
    feat, npatch, tlbr = temporal_merge_nodes(x, tl, temporal_thresh, weighted_avg, head_dim, grid=grid, root_level=root_level,
                                              slow_ver=slow_ver)
    return {"feature": feat, "num_patch": npatch, "tlbr": tlbr}


def cross_frame_node_merging_fast(quadtree_features_video, quadtree_tyxyx_tlbr, temporal_thresh, quadtree_num_patches_per_node,
                                  weighted_avg=False, head_dim=None, quadtree_pos_embs_cos=None, quadtree_pos_embs_sin=None,
                                  pos_emb_weighted_avg=False, *, grid=None, root_level=None):
    """`cross_frame_node_merging_fast` of the reference (token_merging_utils/quadtree_temporal_merger.py:271-287): same positional
    arguments, same return value -- the dict {'feature', 'num_patch', 'tlbr'} of agg_feature_and_metadata (:147-151) -- on the GPU.
    Two keyword arguments are added: `grid = (T, H, W)` and `root_level`, the quadtree partition the nodes come from.
    Differences a caller can meet, each raised instead of answered differently: the list must be sorted by (t, y1, x1) (ValueError),
    `quadtree_num_patches_per_node` must be the box areas (NotImplementedError), no position embeddings here (NotImplementedError)."""
    return _merging_as_the_reference(quadtree_features_video, quadtree_tyxyx_tlbr, temporal_thresh, quadtree_num_patches_per_node,
                                     weighted_avg, head_dim, quadtree_pos_embs_cos, quadtree_pos_embs_sin, grid, root_level, False)


def cross_frame_node_merging_slow(quadtree_features_video, quadtree_tyxyx_tlbr, temporal_thresh, quadtree_num_patches_per_node,
                                  weighted_avg=False, head_dim=None, quadtree_pos_embs_cos=None, quadtree_pos_embs_sin=None,
                                  pos_emb_weighted_avg=False, *, grid=None, root_level=None):
    """`cross_frame_node_merging_slow` (quadtree_temporal_merger.py:289-299): the similarity-sorted pair filter; `head_dim` is ignored
    like the reference ignores it (:293)."""
    return _merging_as_the_reference(quadtree_features_video, quadtree_tyxyx_tlbr, temporal_thresh, quadtree_num_patches_per_node,
                                     weighted_avg, None, quadtree_pos_embs_cos, quadtree_pos_embs_sin, grid, root_level, True)
