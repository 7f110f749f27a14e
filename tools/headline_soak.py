"""One-off soak at the headline size on the GPU box: many seeds x threshold pairs, GPU vs CPU oracle (indices, counts, features)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sttm_oracle as O
from sttm_amd import get_quadtree_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
torch.set_num_threads(32)
tot = exact = 0
worst = 0.0
for (thr, tthr, root) in [(0.85, 0.55, 1), (0.80, 0.50, 1), (0.94, 0.82, 1), (0.85, 0.65, 0)]:
    for seed in range(int(os.environ.get("SEEDS", "20"))):
        x = synth_video(128, 1024, 14, 14, seed=1000 + seed)
        ef, en, et = O.get_quadtree_features(x, thr, tthr, root)
        f, n, t = (o.cpu() for o in get_quadtree_features(x.to(dev), thr, tthr, root))
        ok = t.shape == et.shape and torch.equal(t, et) and torch.equal(n, en)
        tot += 1; exact += int(ok)
        if ok: worst = max(worst, float((f - ef).abs().max()))
        else: print("mismatch", thr, tthr, root, seed, t.shape, et.shape)
print(f"headline-size soak: {exact} of {tot} videos index-exact, worst feature error on those {worst:.1e}")
