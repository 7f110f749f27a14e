"""Secondary configurations of BASELINE.json on one GPU (drop-in API, device-resident inputs): videos/s and keep ratio."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
CONFIGS = [
    ("C1 T=8 14x14x1024 f32 spatial 0.85", 8, 1024, 14, 14, torch.float32, 0.85, -1.0),
    ("C2 T=64 14x14x1024 f32 STTM(0.85,0.65)", 64, 1024, 14, 14, torch.float32, 0.85, 0.65),
    ("C3 T=128 14x14x1024 f32 STTM(0.85,0.55)", 128, 1024, 14, 14, torch.float32, 0.85, 0.55),
    ("C4 T=128 20x36x1024 f32 STTM(0.85,0.60)", 128, 1024, 20, 36, torch.float32, 0.85, 0.60),
    ("C4 T=128 18x26x1024 f32 STTM(0.85,0.60)", 128, 1024, 18, 26, torch.float32, 0.85, 0.60),
    ("C4 T=128 13x24x1024 f32 STTM(0.85,0.60)", 128, 1024, 13, 24, torch.float32, 0.85, 0.60),
    ("C5 T=180 14x14x1024 f32 STTM(0.94,0.82)", 180, 1024, 14, 14, torch.float32, 0.94, 0.82),
    ("real T=128 14x14x3584 bf16 STTM(0.85,0.55)", 128, 3584, 14, 14, torch.bfloat16, 0.85, 0.55),
    ("real T=128 14x14x8192 bf16 STTM(0.85,0.55)", 128, 8192, 14, 14, torch.bfloat16, 0.85, 0.55),
]
for name, T, C, H, W, dt, thr, tthr in CONFIGS:
    pool = [synth_video(T, C, H, W, seed=i, dtype=dt, device=dev, gen_device=dev) for i in range(4)]
    for x in pool:
        f, n, t = get_quadtree_features(x, thr, tthr, 1)
    torch.cuda.synchronize()
    reps = 40
    t0 = time.perf_counter()
    for i in range(reps):
        f, n, t = get_quadtree_features(pool[i % 4], thr, tthr, 1)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / reps
    print(f"{name:48s} {dt_s * 1e6:8.1f} us/video {1 / dt_s:9.1f} videos/s  keep {f.shape[0] / (T * H * W) * 100:5.1f} %")
