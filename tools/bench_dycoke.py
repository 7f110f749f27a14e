"""DyCoke stage-1 pruning on the GPU box: time per video at the headline size and the HBM rate of its kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd.dycoke_merger import dycoke_ttm
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
T, C, P = 128, 1024, 196
pool = [synth_video(T, C, 14, 14, seed=s, device=dev, gen_device=dev).permute(0, 2, 3, 1).reshape(T * P, C).contiguous() for s in range(4)]
for i in range(4): f, idx = dycoke_ttm(pool[i], T, 0.7)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 40
ev[0].record()
for i in range(n): f, idx = dycoke_ttm(pool[i % 4], T, 0.7)
ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / n
n_pairs = T // 2 + (T - 4 + 3) // 4
byt = 2 * n_pairs * P * C * 4 + 2 * f.numel() * 4          # both rows of every pair + gather read/write
print(f"dycoke_ttm T={T} P={P} C={C} fp32 prune 0.7: {us:.1f} us/video = {1e6 / us:.0f} videos/s, keeps {f.shape[0]} of {T * P} tokens; "
      f"{byt / 1e6:.0f} MB algorithmic -> {byt / us / 1e3:.0f} GB/s ({byt / us / 1e3 / 8000 * 100:.0f} % of 8 TB/s)")
