"""One-off soak of the wide 16-bit rows (32-byte packs in the spatial kernel) at full clip length on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sttm_oracle as O
from sttm_amd import get_quadtree_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
torch.set_num_threads(32)
tot = exact = 0
for (C, dt, thr, tthr) in [(3584, torch.bfloat16, 0.85, 0.55), (3584, torch.bfloat16, 0.80, 0.50), (4096, torch.float16, 0.85, 0.55),
                           (2560, torch.bfloat16, 0.90, 0.70)]:
    for seed in range(int(os.environ.get("SEEDS", "6"))):
        x = synth_video(128, C, 14, 14, seed=500 + seed, dtype=dt)
        ef, en, et = O.get_quadtree_features(x, thr, tthr, 1)
        f, n, t = (o.cpu() for o in get_quadtree_features(x.to(dev), thr, tthr, 1))
        ok = t.shape == et.shape and torch.equal(t, et) and torch.equal(n, en) and torch.equal(f, ef)
        tot += 1; exact += int(ok)
        if not ok: print("mismatch", C, dt, thr, tthr, seed, t.shape, et.shape)
print(f"wide-row soak: {exact} of {tot} 128-frame videos bit-identical (indices, counts and 16-bit features)")
