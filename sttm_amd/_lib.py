"""ctypes binding of libsttm_hip.so (the C ABI declared in include/sttm_hip.h).

There is no CPU fallback: if the library is missing or does not load, importing the product path fails
loudly.  Build it with `python -m sttm_amd.build` (hipcc, gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STTM_LIB=dev selects the development build (python -m sttm_amd.build --dev: measurement hooks for tools/, never the product)
# (any other value of STTM_LIB = the FILE NAME of another build inside sttm_amd/lib/ -- same-box A/B of two builds in tools/; only the
#  base name is used, so the variable cannot point the product binding at a shared object outside that directory)
_which = os.path.basename(os.environ.get("STTM_LIB", ""))
LIB_PATH = os.path.join(_HERE, "lib", "libsttm_hip_dev.so" if _which == "dev" else
                        (_which if _which.startswith("libsttm_hip") and _which.endswith(".so") else "libsttm_hip.so"))

STTM_F32, STTM_BF16, STTM_F16 = 0, 1, 2
ERR_ARG, ERR_UNSUPPORTED, ERR_LAUNCH, ERR_INDEX, ERR_PARITY, ERR_TIMEOUT = -1, -2, -3, -4, -5, -6
ABI_VERSION = 8            # STTM_ABI_VERSION of include/sttm_hip.h this binding was written for
CNT_NODES, CNT_CANDIDATES, CNT_EDGES, CNT_OUT, CNT_ITERS, CNT_OVERFLOW, CNT_LEAFNODES, CNT_SLOTS = 0, 1, 2, 3, 4, 5, 6, 8
OVF_BARRIER_TIMEOUT = 64   # STTM_OVF_BARRIER_TIMEOUT
EVENT_SLOTS = 5            # STTM_EVENT_SLOTS
BATCH_MAX = 16             # STTM_BATCH_MAX
EARLY_SLOTS = 64           # STTM_EARLY_SLOTS

# every symbol include/sttm_hip.h declares, with its ctypes signature
_vp, _i, _i64, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    "sttm_abi_version": (_i, []),
    "sttm_last_error": (ctypes.c_char_p, []),
    "sttm_build_tag": (ctypes.c_char_p, []),
    "sttm_quadtree_num_levels": (_i, [_i, _i, _i]),
    "sttm_quadtree_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sttm_quadtree_merge": (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _f, _f, _i, _i, _i, _i,
                                 _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sttm_quadtree_spatial": (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sttm_temporal_merge": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sttm_quadtree_merge_async": (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _f, _f, _i, _i, _i, _i,
                                       _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "sttm_quadtree_merge_batch": (_i, [_i, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _f, _f, _i, _i, _i, _i,
                                       _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "sttm_quadtree_merge_pooled": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i]),
    "sttm_release_streams": (_i, []),
    "sttm_configure": (_i, [ctypes.c_char_p, _i]),
    "sttm_wait_counts": (_i, [_vp, _i, _i]),
    "sttm_quadtree_merge_packed": (_i, [_vp]),
    "sttm_wait_counts_early": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sttm_quadtree_apply": (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "sttm_merge_dst_idx": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sttm_tome_workspace_bytes": (_sz, [_i, _i, _i]),
    "sttm_tome_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sttm_pool2d_out_side": (_i, [_i, _i, _i]),
    "sttm_pool2d": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sttm_dycoke_workspace_bytes": (_sz, [_i, _i, _i]),
    "sttm_dycoke_out_rows": (ctypes.c_int64, [_i, _i, _i]),
    "sttm_dycoke_ttm": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "sttm_octree_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sttm_octree_build": (_i, [_vp, _i, _i, _i, _i, ctypes.c_float, _i, _vp, _sz, _vp, _vp, _vp]),
    "sttm_resize_nearest": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
}



class MergeArgs(ctypes.Structure):
    """sttm_merge_args of include/sttm_hip.h (ABI v6): the argument block of sttm_quadtree_merge_packed."""
    _fields_ = [("x", _vp), ("stride_t", _i64), ("stride_c", _i64), ("stride_h", _i64), ("stride_w", _i64),
                ("T", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("threshold", _f), ("temporal_thresh", _f),
                ("root_level", ctypes.c_int32), ("weighted_avg", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("slow_ver", ctypes.c_int32),
                ("workspace", _vp), ("workspace_bytes", _sz),
                ("feat_out", _vp), ("npatch_out", _vp), ("tlbr_out", _vp), ("counts", _vp),
                ("counts_host", _vp), ("seq", ctypes.c_int32), ("n_early", ctypes.c_int32), ("early_host", _vp),
                ("events", _vp), ("stream", _vp), ("flags", ctypes.c_int32), ("idx_out", _vp)]


_lib = None


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the STTM HIP extension is not built. Run `python -m sttm_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.sttm_abi_version() != ABI_VERSION:
        raise RuntimeError("libsttm_hip.so ABI version mismatch")
    _lib = lib
    return lib


def configure(**kw):
    """Tuning / test switches of the library (sttm_configure, keys in include/sttm_hip.h); none changes results except
    `tome_split` (which match kernel computes the fp32 ToMe scores: differences within the fp32 rounding noise)."""
    lib = load()
    for k, v in kw.items():
        raise_for(lib.sttm_configure(k.encode(), int(v)))


def build_tag():
    return load().sttm_build_tag().decode()


def last_error():
    return load().sttm_last_error().decode("utf-8", "replace")


def raise_for(code):
    """Map a negative return code to the exception type the reference raises in the same situation."""
    if code >= 0:
        return
    msg = last_error()
    if code == ERR_INDEX:
        raise IndexError(msg)                 # size_per_level[root_level] (quadtree_builder.py:111)
    if code == ERR_PARITY:
        raise RuntimeError(msg)               # sumpool on mixed parity (quadtree_spatial_merger.py:63-84)
    if code == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if code == ERR_ARG:
        raise ValueError(msg)
    raise RuntimeError(f"libsttm_hip error {code}: {msg}")


class KernelEvents:
    """STTM_EVENT_SLOTS caller-owned hipEvent_t handles for the `events` argument of sttm_quadtree_merge_async / _batch
    (per-kernel timing: the library records them on the launch stream, the caller reads them).  The HIP entry points are
    resolved through the library's own dependency on libamdhip64, i.e. the runtime instance the kernels run on."""
    NAMES = ("spatial", "pairs", "labels", "group_mean")

    def __init__(self):
        lib = load()
        self._create, self._sync, self._elapsed = lib.hipEventCreate, lib.hipEventSynchronize, lib.hipEventElapsedTime
        self._destroy = lib.hipEventDestroy
        self._create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self._sync.argtypes = [ctypes.c_void_p]
        self._destroy.argtypes = [ctypes.c_void_p]
        self._elapsed.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.handles = (ctypes.c_void_p * EVENT_SLOTS)()
        for i in range(EVENT_SLOTS):
            ev = ctypes.c_void_p()
            if self._create(ctypes.byref(ev)) != 0:
                raise RuntimeError("hipEventCreate failed")
            self.handles[i] = ev

    def pointer(self):
        return ctypes.cast(self.handles, ctypes.c_void_p)

    def elapsed_ms(self):
        """Wait for the last event and return the milliseconds of the four intervals (NAMES)."""
        if self._sync(self.handles[EVENT_SLOTS - 1]) != 0:
            raise RuntimeError("hipEventSynchronize failed")
        out = []
        for i in range(EVENT_SLOTS - 1):
            ms = ctypes.c_float()
            if self._elapsed(ctypes.byref(ms), self.handles[i], self.handles[i + 1]) != 0:
                raise RuntimeError("hipEventElapsedTime failed")
            out.append(ms.value)
        return out

    def __del__(self):
        try:
            for h in self.handles:
                if h:
                    self._destroy(h)
        except Exception:      # noqa: BLE001  -- interpreter shutdown
            pass


class BoundedCache(dict):
    """Per-(device, stream) scratch cache with a bound: a caller that keeps creating streams must not pin one workspace
    (≈100 MB for a 128-frame clip) per stream it ever used.  Oldest entries go first; the tensors are freed by the caching
    allocator once the work queued on them has finished (they were allocated on that stream)."""
    def __init__(self, limit=8):
        super().__init__()
        self.limit = int(os.environ.get("STTM_WS_CACHE", limit))

    def __setitem__(self, key, value):
        if key in self:
            super().__delitem__(key)
        super().__setitem__(key, value)
        while len(self) > self.limit:
            super().__delitem__(next(iter(self)))
