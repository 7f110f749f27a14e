"""Pool fused into the spatial kernel's leaf load against the two-step form (GPU box): us per video, a few shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features, get_quadtree_features_from_pooled_input
from sttm_amd.upstream import get_2dPool
dev = torch.device("cuda:0")


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


SHAPES = ((128, 3584, torch.bfloat16), (128, 1024, torch.float32), (128, 1024, torch.bfloat16), (128, 2048, torch.bfloat16))
if os.environ.get('ONLY'):
    SHAPES = SHAPES[:2]
for T, C, dt in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    base = torch.randn(T // 8, 7, 7, C, device=dev, generator=g).repeat_interleave(8, 0).repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :27, :27]
    src = [(base + 0.3 * torch.randn(T, 27, 27, C, device=dev, generator=g)).reshape(T, 729, C).to(dt).contiguous() for _ in range(3)]
    reps = 24
    n = get_quadtree_features_from_pooled_input(src[0], 0.85, 0.55, 1)[0].shape[0]

    def fused():
        for i in range(reps):
            get_quadtree_features_from_pooled_input(src[i % 3], 0.85, 0.55, 1, force_fused=True)

    def two():
        for i in range(reps):
            p = get_2dPool(src[i % 3], stride=2, mode="bilinear")
            get_quadtree_features(p.reshape(T, 14, 14, C).permute(0, 3, 1, 2), 0.85, 0.55, 1)

    def pool_only():
        for i in range(reps):
            get_2dPool(src[i % 3], stride=2, mode="bilinear")
    eb = src[0].element_size()
    B = eb * C * T * 729 + eb * C * n + 24 * n
    tf, t2, tp = timed(fused, reps), timed(two, reps), timed(pool_only, reps)
    print(f"T={T} C={C} {str(dt).split('.')[-1]}: fused {tf:.1f} us, two-step {t2:.1f} us (pool alone {tp:.1f}), N'={n}, B={B / 1e6:.0f} MB -> fused {B / tf / 1e6:.2f} TB/s", flush=True)
