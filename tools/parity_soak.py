"""Headline-size parity soak (GPU box): N synth-v1 videos (T=128, 14x14x1024 fp32; seeds and threshold pairs cycled) through the BATCH entry
point against the CPU oracle -- merged-token indices / counts bit-exact?  features within 1e-5?  Prints the per-video exact-match rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sttm_oracle as O
from sttm_amd import get_quadtree_features_batch
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "192"))
PAIRS = [(0.85, 0.55), (0.80, 0.50), (0.85, 0.65), (0.94, 0.82), (0.90, 0.60), (0.75, 0.45)]
torch.set_num_threads(min(32, os.cpu_count() or 8))
exact = feat_ok = 0
worst = 0.0
bad = []
t0 = time.perf_counter()
for b0 in range(0, N, 24):
    ids = list(range(b0, min(N, b0 + 24)))
    thr, tthr = PAIRS[(b0 // 24) % len(PAIRS)]
    vids = [synth_video(128, 1024, 14, 14, seed=31000 + i) for i in ids]            # CPU generator: the oracle's own inputs
    outs = get_quadtree_features_batch([v.to(dev) for v in vids], thr, tthr, 1)
    for i, v, (f, n, t) in zip(ids, vids, outs):
        ef, en, et = O.get_quadtree_features(v, thr, tthr, 1)
        ok = t.shape == et.shape and torch.equal(t.cpu(), et) and torch.equal(n.cpu(), en)
        exact += ok
        if ok:
            err = float((f.cpu() - ef).abs().max())
            worst = max(worst, err)
            feat_ok += err <= 1e-5
        else:
            bad.append((i, thr, tthr, int(t.shape[0]), int(et.shape[0])))
    print(f"{min(N, b0 + 24)} videos: index-exact {exact}, features <= 1e-5 {feat_ok}, worst feature error {worst:.2e}, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"RESULT: {exact}/{N} videos index-exact, {feat_ok}/{N} with features within 1e-5 (max {worst:.2e}); mismatching videos: {bad}")
