"""Bandwidth of the upstream pooling kernel on the GPU box (LLaVA-Video shape: T x 27 x 27 x C -> T x 14 x 14 x C)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd.upstream import get_2dPool
dev = torch.device("cuda:0")
for (T, C, dtype, mode) in [(128, 3584, torch.bfloat16, "bilinear"), (128, 1024, torch.float32, "bilinear"), (128, 3584, torch.bfloat16, "average")]:
    pool = [torch.randn(T, 729, C, device=dev).to(dtype) for _ in range(3)]
    for i in range(3): y = get_2dPool(pool[i], 2, mode=mode, num_patches_per_side=27)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 30
    ev[0].record()
    for i in range(n): y = get_2dPool(pool[i % 3], 2, mode=mode, num_patches_per_side=27)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / n
    eb = 2 if dtype != torch.float32 else 4
    byt = (T * 729 * C + y.numel()) * eb
    # the torch formulation the reference runs (permute+contiguous, interpolate / pool, permute+contiguous)
    import torch.nn.functional as F
    def ref(x):
        v = x.view(T, 27, 27, C).permute(0, 3, 1, 2).contiguous()
        v = F.interpolate(v, size=[14, 14], mode="bilinear") if mode == "bilinear" else F.avg_pool2d(v, 2)
        return v.permute(0, 2, 3, 1).reshape(T, -1, C)
    for i in range(3): r = ref(pool[i])
    torch.cuda.synchronize(); ev[0].record()
    for i in range(n): r = ref(pool[i % 3])
    ev[1].record(); torch.cuda.synchronize()
    us_ref = ev[0].elapsed_time(ev[1]) * 1e3 / n
    print(f"pool2d {mode} T={T} C={C} {str(dtype).split('.')[-1]}: {us:.1f} us, {byt / 1e6:.0f} MB algorithmic -> {byt / us / 1e3:.0f} GB/s "
          f"({byt / us / 1e3 / 8000 * 100:.0f} % of 8 TB/s); torch ops of the reference on the same GPU: {us_ref:.1f} us")
