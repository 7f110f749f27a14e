"""Time the decoder hook (slice -> merge -> concat, quadtree_attn_monkey_patch.py:88-117) on the GPU box: the
three-step form (get_quadtree_features + torch.cat) against the fused form (kernels write into the new buffer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features, get_quadtree_features_into, patch_hooks
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
for (T, C, dtype, name) in [(128, 1024, torch.float32, "T=128 14x14x1024 fp32"), (128, 3584, torch.bfloat16, "T=128 14x14x3584 bf16")]:
    pool = []
    for s in range(4):
        vid = synth_video(T, C, 14, 14, seed=s, dtype=dtype, device=dev, gen_device=dev)
        vis = vid.permute(0, 2, 3, 1).reshape(1, T * 196, C)
        hs = torch.cat([torch.randn(1, 14, C, device=dev, dtype=dtype), vis, torch.randn(1, 40, C, device=dev, dtype=dtype)], 1).contiguous()
        pool.append(hs)
    pos = torch.arange(pool[0].shape[1], device=dev).unsqueeze(0)
    for label, into in (("three-step (merge + torch.cat)", None), ("fused (kernels write the new buffer)", get_quadtree_features_into)):
        for it in range(3):
            patch_hooks.quadtree_merge_llava(pool[it % 4], pos, 14, T * 196, T, get_quadtree_features, 0.85, 0.55, 1, False, merge_into_fn=into)
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for it in range(n):
            patch_hooks.quadtree_merge_llava(pool[it % 4], pos, 14, T * 196, T, get_quadtree_features, 0.85, 0.55, 1, False, merge_into_fn=into)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name}: hook {label}: {dt * 1e6:.1f} us")
