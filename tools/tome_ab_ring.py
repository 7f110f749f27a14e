"""A/B of the fp32 ToMe match kernels (PRODUCT build): tome_split 1 (256-tile kernel, two 64 KB stages) against 7 / 8 (the same tile
product from a ring of four / five 32 KB k16 stages), alternating call by call; outputs compared bit for bit first.  Run under
rocprofv3 --kernel-trace --stats (the kernels are different templates: separate rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features, _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
MODES = tuple(int(m) for m in os.environ.get("MODES", "1,7,8").split(","))
for T in (128, 180, 40):
    x = synth_video(T, 1024, 14, 14, seed=3, device=dev, gen_device=dev)
    for ratio in (0.5, 0.85):
        outs = []
        for m in MODES:
            lib.sttm_configure(b"tome_split", m)
            outs.append(get_tome_features(x, ratio, "video"))
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        print(f"T={T} ratio {ratio}: tome_split {MODES} bit-identical: {same}", flush=True)
x = synth_video(128, 1024, 14, 14, seed=3, device=dev, gen_device=dev)
for it in range(20 * len(MODES)):
    lib.sttm_configure(b"tome_split", MODES[it % len(MODES)])
    get_tome_features(x, 0.5, "video")
torch.cuda.synchronize()
lib.sttm_configure(b"tome_split", 1)
