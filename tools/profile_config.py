"""Run one configuration repeatedly (for rocprofv3): T C H W dtype thr tthr root"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features
from sttm_amd.synth import synth_video
T, C, H, W = (int(v) for v in sys.argv[1:5])
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[5]]
thr, tthr, root = float(sys.argv[6]), float(sys.argv[7]), int(sys.argv[8])
dev = torch.device("cuda:0")
pool = [synth_video(T, C, H, W, seed=i, dtype=dt, device=dev, gen_device=dev) for i in range(4)]
for i in range(24):
    get_quadtree_features(pool[i % 4], thr, tthr, root)
torch.cuda.synchronize()
