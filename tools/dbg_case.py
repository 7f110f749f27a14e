"""Debug helper: run one configuration under several library switches and print the counts (GPU box)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.quadtree_interface import quadtree_merge_raw
from sttm_amd.synth import synth_video
from oracle import sttm_oracle as O
T, C, H, W, seed = [int(v) for v in (sys.argv[1:6] + [1024, 32, 14, 14, 20][len(sys.argv) - 1:])][:5]
x = synth_video(T, C, H, W, seed=seed)
ef, en, et = O.get_quadtree_features(x, 0.85, 0.55, 1)
xd = x.to("cuda:0")
for opts in [dict(), dict(fold_labels=1), dict(fold_labels=1, fold_kb=150), dict(force_gmem_labels=1), dict(fold_labels=1, force_gmem_labels=1), dict(pairs_seg=4), dict(pairs_seg=16, pairs_nt=256)]:
    base = dict(fold_labels=0, no_fuse=0, force_gmem_labels=0, fold_kb=64, pairs_seg=0, pairs_nt=0)
    base.update(opts)
    _lib.configure(**base)
    try:
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, n, t, cnt = quadtree_merge_raw(xd, 0.85, 0.55, 1, False, None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        k = cnt[_lib.CNT_OUT]
        ok = k == et.shape[0] and torch.equal(t[:k].cpu(), et)
        print(opts, "counts", cnt, "match", ok, f"{dt*1e3:.2f} ms")
    except Exception as e:
        print(opts, "ERR", str(e)[:200])
