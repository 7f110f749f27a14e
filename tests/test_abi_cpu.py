"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, and its host-side
geometry / argument validation agree with the oracle (no kernel is launched here)."""
import os
import re

import pytest

from sttm_amd import _lib


def _declared_symbols():
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sttm_hip.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sttm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"libsttm_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.sttm_abi_version() == _lib.ABI_VERSION == 8


@pytest.mark.parametrize("H,W", [(14, 14), (27, 27), (20, 36), (18, 26), (13, 24), (16, 22), (10, 30), (7, 7),
                                  (4, 4), (3, 5), (24, 13), (2, 9), (9, 2), (36, 36), (36, 64), (100, 128), (70, 200)])
def test_num_levels_matches_oracle_geometry(H, W):
    from oracle import sttm_oracle as O
    lib = _lib.load()
    n = len(O.level_sizes(H, W))
    for root in range(-n - 1, n + 2):
        got = lib.sttm_quadtree_num_levels(H, W, root)
        try:
            exp = O.Geometry(H, W, root).n_level
        except IndexError:
            assert got == _lib.ERR_INDEX
            continue
        if exp > 6:                                   # kMaxLevels: root cells of up to 32 x 32 leaves
            assert got == _lib.ERR_UNSUPPORTED
        else:
            assert got == exp, (H, W, root)


def test_workspace_bytes_and_errors():
    lib = _lib.load()
    b = lib.sttm_quadtree_workspace_bytes(128, 14, 14, 1024, _lib.STTM_F32, 1)
    assert b >= 128 * 196 * 1024 * 4
    assert lib.sttm_quadtree_workspace_bytes(128, 14, 14, 1024, _lib.STTM_F32, 9) == 0
    assert "root_level" in _lib.last_error()
    with pytest.raises(IndexError):
        _lib.raise_for(_lib.ERR_INDEX)
    with pytest.raises(NotImplementedError):
        _lib.raise_for(_lib.ERR_UNSUPPORTED)


def test_product_path_has_no_cpu_fallback():
    import torch
    from sttm_amd import get_quadtree_features
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_quadtree_features(torch.zeros(2, 8, 14, 14), 0.85)


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sttm_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_abi_version_is_consistent_everywhere():
    """include/sttm_hip.h, the built library, the ctypes binding and __graft_entry__.build() agree on the ABI version."""
    import re
    from sttm_amd import _lib
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sttm_hip.h")).read()
    v = int(re.search(r"#define STTM_ABI_VERSION (\d+)", header).group(1))
    assert v == _lib.ABI_VERSION == _lib.load().sttm_abi_version()
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "_lib.ABI_VERSION" in src


def test_everything_around_the_path_refuses_cpu_tensors():
    """No CPU fallback anywhere in the product package: the producers / baselines around the path fail loudly too."""
    import torch
    from sttm_amd.dycoke_merger import dycoke_ttm
    from sttm_amd.octree_utils import get_octree_features
    from sttm_amd.upstream import get_2dPool, resize_nearest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_2dPool(torch.zeros(2, 81, 8), stride=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        resize_nearest(torch.zeros(2, 81, 8), 9, 9, (5, 5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dycoke_ttm(torch.zeros(6 * 9, 8), 6, 0.7)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_octree_features(torch.zeros(14, 8, 14, 14), 0.85, 0)
    assert get_2dPool(torch.zeros(2, 81, 8), stride=1).shape == (2, 81, 8)          # the reference's identity case needs no device


def test_bounded_cache_evicts_oldest():
    from sttm_amd._lib import BoundedCache
    c = BoundedCache(3)
    for k in range(5):
        c[k] = k * 10
    assert list(c.keys()) == [2, 3, 4]
    c[2] = 99                        # re-inserting refreshes the entry
    c[7] = 70
    assert list(c.keys()) == [4, 2, 7] and c[2] == 99


def test_host_side_size_functions_of_the_baselines():
    lib = _lib.load()
    # pooled side lengths (llava_arch.py:185-192): bilinear rounds up, average / max round down
    assert lib.sttm_pool2d_out_side(27, 2, 2) == 14 and lib.sttm_pool2d_out_side(27, 2, 0) == 13 and lib.sttm_pool2d_out_side(27, 1, 1) == 27
    # DyCoke output rows: T = 128, P = 196, k = 58 -> 33 whole frames + 95 pruned ones
    assert lib.sttm_dycoke_out_rows(128, 196, 58) == 33 * 196 + 95 * 58
    assert lib.sttm_dycoke_out_rows(5, 49, 24) == 2 * 49 + 3 * 24
    # octree: root level outside [2, ..., side] -> no workspace (the wrapper raises IndexError)
    assert lib.sttm_octree_workspace_bytes(1, 14, 32, 0, 9) == 0 and lib.sttm_octree_workspace_bytes(1, 14, 32, 0, 0) > 0
    # ToMe: bad shapes and clips beyond the 2 GiB unit-row matrix of the 256-tile match kernels (32-bit buffer offsets) are refused
    assert lib.sttm_tome_workspace_bytes(35280, 1024, 1) > 0 and lib.sttm_tome_workspace_bytes(1, 1024, 1) == 0
    assert lib.sttm_tome_workspace_bytes(35280, 1024, 3) == 0
    assert lib.sttm_tome_workspace_bytes(1_040_000, 1024, 1) > 0 and lib.sttm_tome_workspace_bytes(1_050_000, 1024, 1) == 0
    assert lib.sttm_tome_workspace_bytes(131_000, 8192, 1) > 0 and lib.sttm_tome_workspace_bytes(131_100, 8192, 1) == 0


def test_configure_accepts_known_keys_only():
    lib = _lib.load()
    assert lib.sttm_configure(b"fold_labels", 0) == 0 and lib.sttm_configure(b"fold_kb", 20) == 0
    assert lib.sttm_configure(b"does_not_exist", 1) == _lib.ERR_ARG
    assert "does_not_exist" in _lib.last_error()
    # switches that make outputs invalid (ablations) are development-build keys: the product library does not know them
    assert lib.sttm_configure(b"col_abl", 1) == _lib.ERR_ARG
    for key in (b"col_walk", b"col_frames", b"col_cap", b"col_pb", b"pair_vec"):
        assert lib.sttm_configure(key, 0) == 0


def test_product_library_has_no_measurement_hooks():
    """wall-clock stamps / ablation modes exist only in the -DSTTM_DEV build (libsttm_hip_dev.so), never in the product library."""
    lib = _lib.load()
    assert not hasattr(lib, "sttm_dev_hooks")
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sttm_amd", "csrc", "api.hip")).read()
    body = src[src.index("int merge_group("):src.index("}  // namespace\n")]
    assert "getenv" not in body          # tuning switches are read once (config()), never per call


def test_feature_file_loaders_refuse_the_cpu(tmp_path):
    import torch
    from sttm_amd.upstream import load_llavavideo_features, load_qwen2vl_features
    p = tmp_path / "v.pt"
    torch.save(torch.zeros(2, 729, 8, dtype=torch.bfloat16), p)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        load_llavavideo_features(str(p), "cpu")
    with pytest.raises(ValueError):
        load_qwen2vl_features(str(p), "cuda:0")              # a [T, tokens, C] file is not the Qwen2-VL [T, H, W, C] format


def test_barrier_timeout_repeats_the_call_on_the_two_launch_label_path():
    """Host logic: a timed-out grid barrier of the fused label stage is reported through its own bit of the overflow slot
    (include/sttm_hip.h: STTM_OVF_BARRIER_TIMEOUT) and the wrapper repeats THAT call with STTM_FLAG_NO_FUSE in its argument
    block; the stream keeps the two-launch path for a bounded number of calls and then re-arms.  No process-wide switch
    (sttm_configure) is touched: other threads' calls are not affected (round-3 review item)."""
    import warnings
    from sttm_amd import quadtree_interface as QI
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sttm_hip.h")).read()
    assert re.search(r"#define\s+STTM_OVF_BARRIER_TIMEOUT\s+64\b", hdr) and _lib.OVF_BARRIER_TIMEOUT == 64
    assert re.search(r"#define\s+STTM_FLAG_NO_FUSE\s+1\b", hdr) and QI.FLAG_NO_FUSE == 1

    class St:                   # the two fields of a stream state the retry logic uses
        no_fuse_left = 0
    st = St()
    calls = []

    def merge(flags):
        calls.append(flags)
        if len(calls) == 1:
            raise QI.BarrierTimeout("timed out")
        return 42
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert QI._with_barrier_retry(st, merge) == 42
    assert calls == [0, QI.FLAG_NO_FUSE] and w and "two-launch" in str(w[0].message)
    assert st.no_fuse_left == QI._NO_FUSE_CALLS
    assert QI._with_barrier_retry(st, merge) == 42 and calls[-1] == QI.FLAG_NO_FUSE and st.no_fuse_left == QI._NO_FUSE_CALLS - 1
    st.no_fuse_left = 0                                   # ... and the stream re-arms the fused path afterwards
    assert QI._with_barrier_retry(st, merge) == 42 and calls[-1] == 0

    def overflow(flags):
        raise RuntimeError("internal list overflow")
    with pytest.raises(RuntimeError):
        QI._with_barrier_retry(st, overflow)            # any other failure is not retried

    def always(flags):
        raise QI.BarrierTimeout("again")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(QI.BarrierTimeout):
            QI._with_barrier_retry(st, always)          # a timeout on the two-launch path itself is not retried again


def test_stream_state_cache_is_lru_and_never_evicts_a_state_in_use(monkeypatch):
    """Round-3 advisor finding: per-(device, stream) state must go out least-recently-USED first (a hit refreshes the entry), a state
    whose lock is held must stay, and an evicted state is parked until its stream has drained instead of being dropped while the
    group-mean kernel may still write its pinned landing pad."""
    import threading
    from sttm_amd import quadtree_interface as QI

    class FakeState:
        def __init__(self, idx, handle):
            self.idx, self.handle, self.lock, self.evicted, self.idle = idx, handle, threading.Lock(), False, False

        def stream_idle(self):
            return self.idle

        def drain(self):
            self.idle = True
    monkeypatch.setattr(QI, "_StreamState", FakeState)
    c = QI._StateCache(3)
    c.limit = 3
    s = [c.get_or_create(0, h) for h in (10, 11, 12)]
    assert c.get_or_create(0, 10) is s[0]                 # a hit: handle 10 is now the most recently used
    c.get_or_create(0, 13)
    assert (0, 11) not in c and (0, 10) in c and len(c) == 3 and s[1].evicted       # 11 was the least recently used
    assert c.graveyard_size() == 1                        # ... and waits for its stream (FakeState.idle is False)
    s[2].lock.acquire()                                   # a thread is inside a call on handle 12 (now the least recently used)
    c.get_or_create(0, 14)
    assert (0, 12) in c and not s[2].evicted and (0, 10) not in c and len(c) == 3
    s[2].lock.release()
    s[1].idle = s[0].idle = True                          # their streams drained: the next retirement sweep lets them go
    c.get_or_create(0, 15)
    assert c.graveyard_size() == 1 and (0, 12) not in c   # only the state evicted just now is still parked


def test_build_tag_depends_on_the_compile_flags():
    from sttm_amd import build
    assert build.source_tag() == build.source_tag(())
    assert build.source_tag(("-DSTTM_DEV",)) != build.source_tag()


def test_argument_block_matches_the_header_and_rejects_null():
    """ABI v5 / v6: the ctypes mirror of sttm_merge_args has the header's fields in the header's order, the packed entry point and
    the early wait validate their arguments without a GPU, and the early wait adds up column words the way the kernels write them."""
    import ctypes
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sttm_hip.h")).read()
    body = re.search(r"typedef struct sttm_merge_args \{(.*?)\} sttm_merge_args;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*$", part.strip())[0])
    assert names == [f[0] for f in _lib.MergeArgs._fields_]
    assert re.search(r"#define\s+STTM_EARLY_SLOTS\s+64\b", hdr) and _lib.EARLY_SLOTS == 64
    lib = _lib.load()
    assert lib.sttm_quadtree_merge_packed(None) == _lib.ERR_ARG
    out = (ctypes.c_int32 * 2)()
    host = (ctypes.c_int32 * 8)()
    early = (ctypes.c_uint64 * 64)()
    assert lib.sttm_wait_counts_early(None, None, 0, 1, 1000, out) == _lib.ERR_ARG
    assert lib.sttm_wait_counts_early(host, early, 65, 1, 1000, out) == _lib.ERR_ARG
    # three columns report for seq 7: survivors 10 + 20 + 5, the last one with the list-overflow flag
    seq = 7
    early[0] = (seq << 32) | 10
    early[1] = (seq << 32) | 20
    early[2] = (seq << 32) | 0x40000000 | 5
    assert lib.sttm_wait_counts_early(host, early, 3, seq, 1000, out) == 0 and (out[0], out[1]) == (35, 1)
    early[2] = (seq << 32) | 0x80000000 | 5                       # the fused label stage's barrier timed out in that column
    assert lib.sttm_wait_counts_early(host, early, 3, seq, 1000, out) == 0 and out[1] == _lib.OVF_BARRIER_TIMEOUT
    early[1] = ((seq - 1) << 32) | 20                             # a stale word: falls back to the classic counts when they arrive
    host[_lib.CNT_OUT], host[_lib.CNT_OVERFLOW], host[_lib.CNT_SLOTS - 1] = 77, 0, seq
    assert lib.sttm_wait_counts_early(host, early, 3, seq, 1000, out) == 0 and (out[0], out[1]) == (77, 0)
    host[_lib.CNT_SLOTS - 1] = seq - 1
    assert lib.sttm_wait_counts_early(host, early, 3, seq, 2000, out) == _lib.ERR_TIMEOUT
