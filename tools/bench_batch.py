"""Batched extension (sttm_quadtree_merge_batch): videos/s against the number of videos per launch set (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features, get_quadtree_features_batch
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
T, C = int(os.environ.get("T", "128")), 1024
pool = [synth_video(T, C, 14, 14, seed=i, device=dev, gen_device=dev) for i in range(16)]
def run(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)
N = 1024
print("one video per call: %.0f videos/s" % run(lambda: [get_quadtree_features(pool[i % 16], 0.85, 0.55, 1) for i in range(N)], N))
for B in (2, 4, 8, 16, 32):
    def f():
        for b0 in range(0, N, B):
            get_quadtree_features_batch([pool[(b0 + k) % 16] for k in range(B)], 0.85, 0.55, 1)
    print(f"batch {B:2d}: {run(f, N):.0f} videos/s")
