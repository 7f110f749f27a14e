"""ToMe `video` r = 0.5 on the GPU box: one call per video against get_tome_features_batch over 1 / 2 / 3 / 4 side streams (videos/s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features, get_tome_features_batch
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
for T, dt in ((128, torch.bfloat16), (180, torch.bfloat16), (128, torch.float32), (180, torch.float32)):
    pool = [synth_video(T, 1024, 14, 14, seed=50 + i, dtype=dt, device=dev, gen_device=dev) for i in range(4)]
    n = 24

    def rate(fn):
        fn(); torch.cuda.synchronize()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = max(best, n / (time.perf_counter() - t0))
        return best
    line = [f"T={T} {str(dt).split('.')[-1]}: one call per video {rate(lambda: [get_tome_features(pool[i % 4], 0.5, 'video') for i in range(n)]):.0f}"]
    for ns in (2, 3, 4):
        line.append(f"{ns} streams {rate(lambda: get_tome_features_batch([pool[i % 4] for i in range(n)], 0.5, 'video', streams=ns)):.0f}")
    print(", ".join(line) + " videos/s", flush=True)
