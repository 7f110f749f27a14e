"""Octree baseline on the GPU box: time per 128-frame video (9 cubes of 14 frames + 2 remainder frames through the quadtree)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd.octree_utils import get_octree_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
T, C = 128, 1024
pool = [synth_video(T, C, 14, 14, seed=s, device=dev, gen_device=dev) for s in range(4)]
for i in range(4): f = get_octree_features(pool[i], 0.85, 1)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 40
ev[0].record()
for i in range(n): f = get_octree_features(pool[i % 4], 0.85, 1)
ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / n
byt = (T * 196 * C + f.numel()) * 4
print(f"get_octree_features T={T} 14x14x{C} fp32 thr 0.85 root 1: {us:.1f} us/video = {1e6 / us:.0f} videos/s, {f.shape[0]} of {T * 196} tokens; "
      f"read-once + write-once bytes {byt / 1e6:.0f} MB -> {byt / us / 1e3:.0f} GB/s ({byt / us / 1e3 / 8000 * 100:.0f} % of 8 TB/s)")
