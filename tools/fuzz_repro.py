"""Re-run ONE case of tests/test_hip_fuzz.py (picked by its pytest id) and print where the device result and the oracle differ.
STTM_LIB=dev + K1_VAR=1 runs the general spatial body instead.   usage: STTM_FUZZ_N=3000 STTM_FUZZ_SEED=7171 STTM_FUZZ_TMAX=12 python tools/fuzz_repro.py T8_C256_24x27_r1_synth"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_fuzz import _cases
from oracle import sttm_oracle as O
from sttm_amd import _lib, get_quadtree_features
from sttm_amd.synth import iid_video, synth_video
want = sys.argv[1]
cases = _cases(int(os.environ.get("STTM_FUZZ_N", "200")), int(os.environ.get("STTM_FUZZ_SEED", "1234")))
if os.environ.get("K1_VAR"):
    _lib.configure(k1_var=int(os.environ["K1_VAR"]))
for c in cases:
    if "T%d_C%d_%dx%d_r%d_%s" % (c[0], c[1], c[2], c[3], c[5], c[10]) != want:
        continue
    T, C, H, W, dtype, root, thr, tthr, weighted, slow, kind, seed = c
    print("case", c)
    x = iid_video(T, C, H, W, seed=seed, dtype=dtype) if kind == "iid" else synth_video(T, C, H, W, seed=seed, dtype=dtype, **(dict(c=0.15, p_static=0.7) if kind == "smooth" else {}))
    ef, en, et = O.get_quadtree_features(x, thr, tthr, root, weighted, slow_ver=slow)
    f, n, t = (o.cpu() for o in get_quadtree_features(x.to("cuda:0"), thr, tthr, root, weighted, slow_ver=slow))
    print("oracle", tuple(et.shape), "device", tuple(t.shape), "tlbr equal", t.shape == et.shape and bool(torch.equal(t, et)))
    so = set(map(tuple, et.tolist())); sd = set(map(tuple, t.tolist()))
    print("only in oracle:", sorted(so - sd)[:12]); print("only on device:", sorted(sd - so)[:12])
    # spatial stage alone
    ef2, en2, et2 = O.get_quadtree_features(x, thr, -1.0, root, weighted)
    f2, n2, t2 = (o.cpu() for o in get_quadtree_features(x.to("cuda:0"), thr, -1.0, root, weighted))
    so = set(map(tuple, et2.tolist())); sd = set(map(tuple, t2.tolist()))
    print("spatial only: equal", so == sd, "only in oracle:", sorted(so - sd)[:12], "only on device:", sorted(sd - so)[:12])
    if t.shape == et.shape and torch.equal(t, et):
        err = (f.float() - ef.float()).abs()
        print("max feature err", float(err.max()), "rows with err", int((err.amax(1) > 0).sum()))
