"""Seeded fuzz of the HIP path against the oracle over random shapes / thresholds / variants (small sizes).
STTM_FUZZ_N / STTM_FUZZ_SEED / STTM_FUZZ_TMAX / STTM_FUZZ_HMAX / STTM_FUZZ_WMAX widen or move the sweep for one-off runs (defaults: 200
cases, seed 1234, grids up to 30 x 40; 70 x 70 reaches the 6-level trees of the split spatial stage)."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        H, W = rng.randint(3, int(os.environ.get("STTM_FUZZ_HMAX", "30"))), rng.randint(3, int(os.environ.get("STTM_FUZZ_WMAX", "40")))
        T = rng.randint(1, int(os.environ.get("STTM_FUZZ_TMAX", "7")))
        C = rng.choice([8, 12, 16, 20, 32, 64, 100, 128, 256])
        dtype = rng.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
        if dtype != torch.float32 and C % 2:
            continue
        root = rng.choice([-2, -1, 0, 1, 1, 1, 2])
        thr = rng.choice([0.6, 0.75, 0.8, 0.85, 0.9, 0.94, 0.95])          # 0.85 / 0.94: the run_vidqa.sh presets
        tthr = rng.choice([-1.0, 0.3, 0.5, 0.55, 0.65, 0.7, 0.82])           # 0.55 / 0.65 / 0.82 likewise
        weighted = rng.random() < 0.25
        slow = rng.random() < 0.2 and tthr > 0
        kind = rng.choice(["synth", "smooth", "iid"])
        out.append((T, C, H, W, dtype, root, thr, tthr, weighted, slow, kind, rng.randint(0, 10 ** 6)))
    return out


@pytest.mark.parametrize("case", _cases(int(os.environ.get("STTM_FUZZ_N", "200")), int(os.environ.get("STTM_FUZZ_SEED", "1234"))), ids=lambda c: "T%d_C%d_%dx%d_r%d_%s" % (c[0], c[1], c[2], c[3], c[5], c[10]))
def test_fuzz_against_oracle(case):
    from oracle import sttm_oracle as O
    from sttm_amd import get_quadtree_features
    from sttm_amd.synth import iid_video, synth_video
    T, C, H, W, dtype, root, thr, tthr, weighted, slow, kind, seed = case
    if kind == "iid":
        x = iid_video(T, C, H, W, seed=seed, dtype=dtype)
    else:
        kw = dict(c=0.15, p_static=0.7) if kind == "smooth" else {}
        x = synth_video(T, C, H, W, seed=seed, dtype=dtype, **kw)
    dev = torch.device("cuda:0")
    try:
        exp = O.get_quadtree_features(x, thr, tthr, root, weighted, slow_ver=slow)
    except (IndexError, RuntimeError) as e:      # out-of-range root level / sum-pool on mixed parity: same error type here
        with pytest.raises(type(e)):
            get_quadtree_features(x.to(dev), thr, tthr, root, weighted, slow_ver=slow)
        return
    try:
        out = get_quadtree_features(x.to(dev), thr, tthr, root, weighted, slow_ver=slow)
    except NotImplementedError as e:             # documented device limits (trees deeper than 6 levels: not reachable with W <= 40)
        pytest.skip(str(e))
    f, n, t = (o.cpu() for o in out)
    ef, en, et = exp
    assert t.shape == et.shape and torch.equal(t, et), f"tlbr differs: {t.shape} vs {et.shape}"
    assert torch.equal(n, en)
    err = (f.float() - ef.float()).abs()
    if dtype == torch.float32:
        assert float(err.max()) <= 1e-5
    else:
        assert float((err / ef.float().abs().clamp_min(1.0)).max()) <= 2 ** -7


@pytest.mark.parametrize("chunk", range(6))
def test_fuzz_through_the_batch_entry_point(chunk):
    """The same random cases (every 3rd one: shapes, dtypes, root levels, weighted / slow variants mixed) through
    get_quadtree_features_batch -- several differently shaped videos per call, each shape twice, so launch sets of several videos form --
    must equal the one-video calls bit for bit (launch sets run 8 group-mean workgroups per frame and 256-thread label columns: neither may
    change a result)."""
    from sttm_amd import get_quadtree_features, get_quadtree_features_batch
    from sttm_amd.synth import iid_video, synth_video
    dev = torch.device("cuda:0")
    cases = _cases(180, 4321)[chunk::6][::1][:30]
    by_params = {}
    for c in cases:
        by_params.setdefault((c[5], c[6], c[7], c[8], c[9]), []).append(c)          # one batch call per (root, thr, tthr, weighted, slow)
    for (root, thr, tthr, weighted, slow), group in by_params.items():
        vids = []
        for (T, C, H, W, dtype, _, _, _, _, _, kind, seed) in group:
            for s in (seed, seed + 1):
                if kind == "iid":
                    vids.append(iid_video(T, C, H, W, seed=s, dtype=dtype).to(dev))
                else:
                    kw = dict(c=0.15, p_static=0.7) if kind == "smooth" else {}
                    vids.append(synth_video(T, C, H, W, seed=s, dtype=dtype, **kw).to(dev))
        try:
            single = [get_quadtree_features(v, thr, tthr, root, weighted, slow_ver=slow) for v in vids]
        except (IndexError, RuntimeError, NotImplementedError) as e:
            with pytest.raises(type(e)):
                get_quadtree_features_batch(vids, thr, tthr, root, weighted, slow_ver=slow)
            continue
        batch = get_quadtree_features_batch(vids, thr, tthr, root, weighted, slow_ver=slow)
        torch.cuda.synchronize()
        for k, (b, s_) in enumerate(zip(batch, single)):
            for u, v in zip(b, s_):
                assert u.shape == v.shape and torch.equal(u.view(torch.uint8), v.view(torch.uint8)), (root, thr, tthr, weighted, slow, k)
