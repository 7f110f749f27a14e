"""ToMe baseline timing on the GPU box: per-kernel times of one sttm_tome_step at the headline size and the
MFMA roofline of the fused match kernel (fp32-input MFMA peak 157.3 TFLOP/s)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
T, C = int(os.environ.get("T", "128")), 1024
x = synth_video(T, C, 14, 14, seed=3, device=dev, gen_device=dev)
if os.environ.get("DTYPE"):                      # DTYPE=bfloat16 / float16: the one-plane match kernels
    x = x.to(getattr(torch, os.environ["DTYPE"]))
if os.environ.get("CONST_INPUT"):                # every element 1: same instructions and counters, (almost) no operand toggling -- the DVFS probe
    x = torch.ones_like(x)
for ratio in (0.5, 0.7, 0.85):
    get_tome_features(x, ratio, "video")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_it = int(os.environ.get("N_IT", "5"))
    for _ in range(n_it):
        f, i = get_tome_features(x, ratio, "video")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    n = T * 196
    tgt = math.ceil(n * (1 - ratio))
    flops, cur = 0, n
    first = True
    while first or cur > tgt:
        first = False
        r = min(cur - tgt, cur // 2)
        flops += 2 * ((cur + 1) // 2) * (cur // 2) * C
        cur -= r
    print(f"ToMe video ratio={ratio}: {dt * 1e3:.2f} ms / video = {1 / dt:.1f} videos/s; {flops / 1e9:.1f} GFLOP matched -> "
          f"{flops / dt / 1e12:.1f} TFLOP/s whole step ({flops / dt / 157.3e12 * 100:.1f} % of the fp32 MFMA peak); out {tuple(f.shape)}")
