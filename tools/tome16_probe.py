"""16-bit ToMe timing on the GPU box: whole get_tome_features per video at T frames (default 128), bf16 and fp16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
T, C = int(os.environ.get("T", "128")), int(os.environ.get("C", "1024"))
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    x = synth_video(T, C, 14, 14, seed=3, device=dev, gen_device=dev).to(dtype)
    for ratio in (0.5, 0.85):
        get_tome_features(x, ratio, "video")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            get_tome_features(x, ratio, "video")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{dtype} T={T} C={C} ratio={ratio}: {dt * 1e3:.3f} ms / video", flush=True)
