"""synth-v1: the synthetic video-token distribution the benchmark and the parity tests run on.

This is the build's own generator; it contains no reference code.  It is
calibrated so that the reference keeps ~62 % of the tokens at spatial threshold 0.85 and ~45 % at
(0.85, 0.55), the budgets `scripts/eval/run_vidqa.sh:54-90` of the reference are tuned for.

The tensor is produced on the CPU generator (seed = video id) as logical [T, C, H, W] and returned as
the *production layout*: a [T, C, H, W] view of [T, H, W, C] memory, exactly what the reference's
patched forward hands to `get_quadtree_features`
(`token_merging_monkey_patch/quadtree_attn_monkey_patch.py:98`).
"""
import torch
import torch.nn.functional as F


def synth_video(T, C=1024, H=14, W=14, seed=0, a=0.5, c=0.4, p_static=0.3, rho_hi=0.98, rho_lo=0.1,
                dtype=torch.float32, device="cpu", gen_device="cpu"):
    """gen_device="cpu" (default) draws from the CPU generator, so a (seed, shape) pair names the same
    tensor everywhere (parity tests, golden vectors).  gen_device="cuda:N" draws the same distribution
    from the device generator instead -- a different sample, only for filling benchmark pools quickly."""
    gd = torch.device(gen_device)
    g = torch.Generator(device=gd).manual_seed(int(seed))
    kw = dict(generator=g, device=gd)
    m = torch.randn(1, C, 1, 1, **kw)

    def field():
        f = 0
        for s in (2, 4, 7, 14):
            f = f + F.interpolate(torch.randn(1, C, s, s, **kw), size=(H, W), mode="nearest") * 0.5
        return f[0]

    x = torch.empty(T, C, H, W, device=gd)
    cur = field()
    for t in range(T):
        if t > 0:
            stat = (torch.rand(1, 1, 4, 4, **kw) < p_static).float()
            rho = F.interpolate(stat * rho_hi + (1 - stat) * rho_lo, size=(H, W), mode="nearest")[0]
            cur = rho * cur + (1 - rho * rho).sqrt() * field()
        x[t] = cur
    x = a * m + x + c * torch.randn(T, C, H, W, **kw)
    # production layout: [T,H,W,C] memory, viewed as [T,C,H,W]
    x = x.permute(0, 2, 3, 1).contiguous().to(dtype)
    if torch.device(device) != x.device:
        x = x.to(device)
    return x.permute(0, 3, 1, 2)


def iid_video(T, C, H, W, seed=0, dtype=torch.float32, device="cpu"):
    """iid-normal tokens in the production layout (almost nothing merges: a worst case for N)."""
    g = torch.Generator().manual_seed(int(seed))
    x = torch.randn(T, H, W, C, generator=g).to(dtype)
    if device != "cpu":
        x = x.to(device)
    return x.permute(0, 3, 1, 2)
