"""GPU parity of the column-walk spatial stage (csrc/spatial_col.inc, round 6): one workgroup per (root cell, chunk of frames) that
also runs the pair stage of quadtree_temporal_merger.py:36-45, 58-73 on node rows kept in LDS.  It is opt-in (`col_walk=1`: launch sets of
several videos -- measured slower than the spatial + pair kernels); `col_walk=2` forces it for one-video calls so that it can be checked against the oracle and against the
spatial + pair kernels of the one-video path (bit-identical outputs: same partial sums in the same order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5
BF16_TOL = 2 ** -7
DEFAULTS = dict(col_walk=0, col_frames=8, col_cap=0, col_pb=0)


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _equal(a, b, what):
    (f, n, t), (ef, en, et) = a, b
    assert t.shape == et.shape, f"{what}: N'={t.shape[0]} vs {et.shape[0]}"
    assert torch.equal(t, et), f"{what}: tlbr differ"
    assert torch.equal(n, en), f"{what}: num_patches differ"
    assert torch.equal(f, ef), f"{what}: features differ (max {float((f.float() - ef.float()).abs().max()):.3e})"


# (T, C, H, W, seed, dtype, thr, tthr, root, weighted, synth kwargs)
CASES = [
    (8, 1024, 14, 14, 300, torch.float32, 0.85, 0.55, 1, False, {}),              # one chunk
    (9, 1024, 14, 14, 301, torch.float32, 0.85, 0.55, 1, False, {}),              # a second chunk of one frame (warm-up + 1)
    (2, 256, 14, 14, 302, torch.float32, 0.85, 0.55, 1, False, {}),
    (33, 256, 14, 14, 303, torch.float32, 0.80, 0.50, 1, False, {}),              # varied tree structure (73-115 nodes per frame)
    (40, 128, 14, 14, 304, torch.float32, 0.80, 0.50, 1, False, dict(c=0.15, p_static=0.7)),   # smooth: long chains, whole root cells merged
    (24, 512, 14, 14, 305, torch.float32, 0.95, 0.90, 1, False, {}),              # nearly everything splits to leaves: 16-node root cells
    (17, 64, 13, 24, 306, torch.float32, 0.85, 0.60, 1, False, {}),               # odd x even grid: alias cells, lone first row
    (12, 96, 7, 7, 307, torch.float32, 0.85, 0.55, 0, False, {}),
    (20, 256, 14, 14, 308, torch.float32, 0.85, 0.55, 1, True, {}),               # weighted_avg (sum-pool pyramid)
    (19, 3584, 14, 14, 309, torch.bfloat16, 0.85, 0.55, 1, False, {}),            # production width: 448 lanes, 7 waves
    (16, 1024, 14, 14, 310, torch.float16, 0.85, 0.55, 1, False, {}),
    (16, 1000, 14, 14, 311, torch.float32, 0.85, 0.55, 1, False, {}),             # idle lanes in the last wave
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "T%d_C%d_%dx%d_s%d" % c[:5])
@pytest.mark.parametrize("knobs", [dict(), dict(col_frames=3), dict(col_cap=2, col_pb=3), dict(col_frames=1, col_cap=1, col_pb=1), dict(col_frames=2, col_cap=3, col_pb=16)],
                         ids=lambda k: "+".join(f"{a}={b}" for a, b in k.items()) or "default")
def test_column_walk_against_oracle_and_the_pair_kernel(case, knobs):
    from oracle import sttm_oracle as O
    from sttm_amd import _lib, get_quadtree_features
    from sttm_amd.synth import synth_video
    T, C, H, W, seed, dtype, thr, tthr, root, weighted, kw = case
    x = synth_video(T, C, H, W, seed=seed, dtype=dtype, **kw)
    exp = O.get_quadtree_features(x, thr, tthr, root, weighted)
    xd = x.to(_dev())
    try:
        _lib.configure(col_walk=0)
        ref = get_quadtree_features(xd, thr, tthr, root, weighted)
        _lib.configure(col_walk=2, **knobs)
        out = get_quadtree_features(xd, thr, tthr, root, weighted)
        torch.cuda.synchronize()
    finally:
        _lib.configure(**DEFAULTS)
    _equal(out, ref, f"{case[:5]} {knobs}: column walk vs spatial + pair kernels")
    f, n, t = out
    ef, en, et = exp
    assert torch.equal(t.cpu(), et) and torch.equal(n.cpu(), en), f"{case[:5]}: indices differ from the oracle"
    err = (f.cpu().float() - ef.float()).abs()
    scale = ef.float().abs().clamp_min(1.0) if dtype != torch.float32 else 1.0
    assert float((err / scale).max()) <= (FP32_TOL if dtype == torch.float32 else BF16_TOL)


def test_launch_sets_on_the_column_walk_equal_one_video_calls():
    """The batch entry point with col_walk=1 (launch sets of several videos: column walk) against one-video calls (spatial + pair kernels)."""
    from sttm_amd import get_quadtree_features, get_quadtree_features_batch
    from sttm_amd.synth import synth_video
    vids = [synth_video(T, 512, 14, 14, seed=400 + i).to(_dev()) for i, T in enumerate([24] * 20 + [9, 17, 24, 24])]
    from sttm_amd import _lib
    single = [get_quadtree_features(v, 0.80, 0.50, 1) for v in vids]
    try:
        _lib.configure(col_walk=1)
        batch = get_quadtree_features_batch(vids, 0.80, 0.50, 1)
        torch.cuda.synchronize()
    finally:
        _lib.configure(**DEFAULTS)
    for i, (a, b) in enumerate(zip(batch, single)):
        _equal(a, b, f"video {i}")


def test_headline_size_column_walk_equals_the_pair_kernel():
    from sttm_amd import _lib, get_quadtree_features
    from sttm_amd.synth import synth_video
    x = synth_video(128, 1024, 14, 14, seed=7).to(_dev())
    try:
        _lib.configure(col_walk=0)
        ref = get_quadtree_features(x, 0.85, 0.55, 1)
        _lib.configure(col_walk=2)
        out = get_quadtree_features(x, 0.85, 0.55, 1)
        torch.cuda.synchronize()
    finally:
        _lib.configure(**DEFAULTS)
    _equal(out, ref, "headline")
