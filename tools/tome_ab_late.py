"""A/B of the 256-tile ToMe match kernel (DEVELOPMENT build): STTM_TOME_ABL = 0 (the product form: branch-free running max, one-plane
kernels read step n+1's fragments after four MFMAs of step n) against 3 (the round-2 form) and 4 (reads after the first MFMA),
alternating call by call so that clock drift hits all alike; outputs of the three are compared bit for bit first.  Run under
rocprofv3 --kernel-trace --stats: the forms are different template instances and show up as separate rows."""
import os, sys
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
x = synth_video(128, 1024, 14, 14, seed=3, device=dev, gen_device=dev)
MODES = tuple(os.environ.get("MODES", "0,3,4").split(","))      # MODES=0,5,6,1: the ablation ladder (fp32 only, outputs invalid)
LADDER = any(m in ("1", "2", "5", "6") for m in MODES)
for xin in ((x,) if LADDER else (x, x.to(torch.bfloat16), x.to(torch.float16))):
    for ratio in (() if LADDER else (0.5, 0.85)):
        outs = []
        for m in MODES:
            os.environ["STTM_TOME_ABL"] = m
            outs.append(get_tome_features(xin, ratio, "video"))
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        print(f"{xin.dtype} ratio {ratio}: modes {MODES} bit-identical: {same}", flush=True)
    for it in range(60):
        os.environ["STTM_TOME_ABL"] = MODES[it % len(MODES)]
        get_tome_features(xin, 0.5, "video")
    torch.cuda.synchronize()
