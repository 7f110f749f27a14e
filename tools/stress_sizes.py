"""One-off size stress on the GPU box: long clips (columns spill to global scratch), big grids, wide channels -- each against
the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sttm_oracle as O
from sttm_amd import get_quadtree_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
CASES = [  # T, C, H, W, dtype, thr, tthr, root
    (600, 64, 14, 14, torch.float32, 0.85, 0.55, 1),       # 9600 slots per column -> global-memory label path
    (1024, 32, 14, 14, torch.float32, 0.85, 0.55, 1),
    (300, 128, 27, 27, torch.float32, 0.85, 0.60, 1),
    (4, 64, 64, 64, torch.float32, 0.85, 0.55, 2),
    (16, 8192, 14, 14, torch.bfloat16, 0.85, 0.55, 1),
    (32, 4096, 14, 14, torch.float32, 0.85, 0.55, 1),
    (200, 96, 20, 36, torch.float16, 0.85, 0.60, 0),
]
for (T, C, H, W, dt, thr, tthr, root) in CASES:
    x = synth_video(T, C, H, W, seed=T + C, dtype=dt)
    t0 = time.time()
    try:
        ef, en, et = O.get_quadtree_features(x, thr, tthr, root)
    except Exception as e:
        print("oracle raised", type(e).__name__, e); continue
    t1 = time.time()
    try:
        f, n, t = (o.cpu() for o in get_quadtree_features(x.to(dev), thr, tthr, root))
    except NotImplementedError as e:
        print(f"T={T} C={C} {H}x{W} {dt}: device limit: {e}"); continue
    ok_idx = torch.equal(t, et) and torch.equal(n, en)
    err = float((f.float() - ef.float()).abs().max()) if f.shape == ef.shape else float("nan")
    print(f"T={T} C={C} {H}x{W} {str(dt).split('.')[-1]} root={root}: {et.shape[0]} tokens, index-exact={ok_idx}, max feature err={err:.2e} (oracle {t1 - t0:.1f} s)")
