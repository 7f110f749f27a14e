"""Calibration, not product: what the vendor library's GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on THIS box at the ToMe match
shape -- a @ b^T with a, b = [12544, 1024] unit rows (first iteration at T = 128) and [17640, 1024] (T = 180) -- so that the fused match
kernel's MfmaUtil can be read against a plain GEMM that also has to WRITE its n x n scores (which the fused kernel never does).
Synthetic unit rows of the clip the ToMe benches use.  DTYPE = bfloat16 | float16; prints one line per shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
dt = getattr(torch, os.environ.get("DTYPE", "bfloat16"))
n_it = int(os.environ.get("N_IT", "6"))
for T in (int(t) for t in os.environ.get("TS", "128,180").split(",")):
    x = synth_video(T, 1024, 14, 14, seed=3, device=dev, gen_device=dev).reshape(-1, 1024)
    x = torch.nn.functional.normalize(x, dim=-1).to(dt)
    a, b = x[0::2].contiguous(), x[1::2].contiguous()
    out = torch.empty(a.shape[0], b.shape[0], dtype=dt, device=dev)
    torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it):
        torch.matmul(a, b.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n_it
    flop = 2.0 * a.shape[0] * b.shape[0] * 1024
    print(f"library GEMM {os.environ.get('DTYPE', 'bfloat16')} T={T}: [{a.shape[0]} x 1024] @ [1024 x {b.shape[0]}] {us:.0f} us = {flop / us / 1e6:.0f} TFLOP/s "
          f"(writes {out.numel() * out.element_size() / 1e6:.0f} MB of scores)")
