"""Triage helper for the GPU box: run a few small cases through the HIP path and print where they
diverge from the oracle (counts first, then rows)."""
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sttm_oracle as O
from sttm_amd.quadtree_interface import quadtree_merge_raw
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
CASES = [
    (2, 64, 14, 14, 0.85, -1.0, 1, torch.float32),
    (4, 64, 14, 14, 0.85, 0.55, 1, torch.float32),
    (4, 1024, 14, 14, 0.85, 0.55, 1, torch.float32),
    (3, 64, 20, 36, 0.85, 0.60, 1, torch.float32),
    (3, 64, 27, 27, 0.80, 0.50, 0, torch.float32),
    (3, 64, 14, 14, 0.85, 0.55, -1, torch.float32),
    (3, 64, 14, 14, 0.85, 0.55, 2, torch.float32),
    (4, 64, 14, 14, 0.85, 0.55, 1, torch.bfloat16),
]
for (T, C, H, W, thr, tthr, root, dt) in CASES:
    x = synth_video(T, C, H, W, seed=7, dtype=dt)
    ef, en, et, dbg = O.get_quadtree_features(x, thr, tthr, root, return_debug=True)
    t0 = time.time()
    f, n, t, cnt = quadtree_merge_raw(x.to(dev), thr, tthr, root, False, None)
    dt_ms = (time.time() - t0) * 1e3
    nout = cnt[3]
    t, n, f = t[:nout].cpu(), n[:nout].cpu(), f[:nout].cpu()
    ok_idx = t.shape == et.shape and torch.equal(t, et) and torch.equal(n, en)
    err = float((f.float() - ef.float()).abs().max()) if ok_idx else float("nan")
    print(f"case T={T} C={C} {H}x{W} root={root} {dt}: counts={cnt[:6]} expectN={dbg['spatial_tlbr'].shape[0]} "
          f"expectL={None if dbg.get('candidates') is None else dbg['candidates'].shape[0]} "
          f"expectL'={None if dbg.get('kept') is None else dbg['kept'].shape[0]} expectN'={et.shape[0]} "
          f"index_ok={ok_idx} feat_err={err:.2e} ({dt_ms:.1f} ms)")
    if not ok_idx:
        m = min(t.shape[0], et.shape[0])
        bad = (t[:m] != et[:m]).any(1).nonzero().flatten()
        if len(bad):
            b = int(bad[0])
            print("   first bad row", b, t[b].tolist(), "expected", et[b].tolist())
