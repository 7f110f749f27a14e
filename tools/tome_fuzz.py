"""ToMe fuzz on the GPU box: random (T, C, ratio, heads) clips through every fp32 match kernel and both work splits of the 256-tile
kernels against the CPU oracle (the checker; oracle/sttm_oracle.py), kept-token ids as sets and features as (id -> row) maps, the
bars of tests/test_hip_parity.py::_compare_tome.  usage: python tools/tome_fuzz.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sttm_oracle as O
from sttm_amd import get_tome_features, _lib
from sttm_amd.synth import synth_video
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
TOL = 1e-5
exact = near = 0
worst = 0.0
try:
    for k in range(n_cases):
        T = rng.choice([1, 2, 3, 5, 8, 13, 20, 28])
        C = rng.choice([64, 96, 128, 250, 256, 512, 1000, 1024])
        n_head = rng.choice([1, 1, 1, 2, 4]) if C % 4 == 0 else 1
        ratio = rng.choice([0.3, 0.5, 0.6, 0.7, 0.85, 0.9])
        split, flat = rng.choice([1, 2, 3, 4, 6, 7, 0]), rng.choice([0, 1, 2])
        x = synth_video(T, C, 14, 14, seed=9000 + k)
        ef, ei = O.get_tome_features(x, ratio, "video", n_head)
        _lib.configure(tome_split=split, tome_flat=flat)
        f, i = get_tome_features(x.to(dev), ratio, "video", n_head)
        f, i = f.cpu(), i.cpu()
        go, xo = torch.argsort(i), torch.argsort(ei)
        gi, gf, xi, xf = i[go], f[go], ei[xo], ef[xo]
        what = f"case {k}: T={T} C={C} heads={n_head} ratio={ratio} tome_split={split} tome_flat={flat}"
        if torch.equal(gi, xi):
            rows = (gf - xf).abs().amax(dim=1)
            bad = (rows > TOL).nonzero().flatten()
            if bad.numel() == 0:
                worst = max(worst, float(rows.max()))
                exact += 1
            else:
                # an argmax near-tie (two b candidates within the fp32 summation noise of the CPU matmul) sends ONE source to the
                # other candidate: the kept set is the same and the destination rows of the swapped sources differ.  The referee is the
                # oracle in float64: whoever it agrees with resolved the near-tie by the true scores (the split-plane kernels' products
                # are exact, so it is usually them: case 13 of seed 11 -- all seven split forms equal the float64 oracle, the fp32 CPU
                # matmul is the one that differs from it in 3 rows)
                f64, i64 = O.get_tome_features(x.double(), ratio, "video", n_head)
                o64 = torch.argsort(i64)
                vs64 = ((gf - f64[o64].float()).abs().amax(dim=1) > TOL).sum().item() if torch.equal(gi, i64[o64]) else -1
                print(f"{what}: same ids, {bad.numel()} rows differ from the fp32 oracle (ids {gi[bad][:6].tolist()}), {vs64} from the float64 oracle: argmax near-tie", flush=True)
                assert vs64 == 0 or bad.numel() <= max(2, len(gi) // 1000), what
                near += 1
        else:
            both = sorted(set(gi.tolist()) & set(xi.tolist()))
            agree = len(both) / len(xi)
            print(f"{what}: near-tie, id agreement {agree:.5f}", flush=True)
            assert agree >= 0.999, what
            near += 1
finally:
    _lib.configure(tome_split=2, tome_flat=1)
print(f"tome fuzz: {n_cases} cases, {exact} id-exact (max feature err {worst:.2e}), {near} with a near-tie swap (>= 99.9 % ids)")
