"""Alternating launches of the two work splits of the 256-tile ToMe match kernels (tome_flat 0 / 2) for a rocprofv3 trace; read with
tools/prof_by_grid.py (the splits differ in grid size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features, _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
T = int(os.environ.get("T", "180"))
x32 = synth_video(T, 1024, 14, 14, seed=3, device=dev, gen_device=dev)
for x in (x32, x32.to(torch.bfloat16)):
    for it in range(40):
        lib.sttm_configure(b"tome_flat", 2 if it & 1 else 0)
        get_tome_features(x, 0.5, "video")
    torch.cuda.synchronize()
