#!/usr/bin/env python3
"""Golden vectors for the octree baseline, made by RUNNING THE REFERENCE's `get_octree_features`
(token_merging_utils/octree_utils.py:293-386) in the build container (CPU).  Only data is written.

    python tests/golden/make_golden_octree.py        # rewrites tests/golden/oct_*.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("STTM_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from sttm_amd.synth import synth_video                                   # noqa: E402
from token_merging_utils.octree_utils import get_octree_features        # noqa: E402  (reference)

CASES = [
    dict(name="oct_t14_s14_r0", T=14, side=14, C=32, thr=0.85, root=0, seed=1),              # one cube, 4 levels, odd sizes (7)
    dict(name="oct_t30_s14_r1", T=30, side=14, C=32, thr=0.85, root=1, seed=2),              # two cubes + 2 remainder frames
    dict(name="oct_t16_s8_r0", T=16, side=8, C=16, thr=0.80, root=0, seed=3),                # all-even pyramid 8-4-2
    dict(name="oct_t9_s7_r0_smooth", T=9, side=7, C=24, thr=0.75, root=0, seed=4, smooth=True),   # 7-4-2, remainder 2
    dict(name="oct_t5_s14_short", T=5, side=14, C=16, thr=0.85, root=1, seed=5),             # T < side: per-frame quadtree only
    dict(name="oct_t20_s10_r2", T=20, side=10, C=16, thr=0.6, root=2, seed=6, smooth=True),  # 10-5-3-2, root at level 2
    dict(name="oct_t14_s14_r1_c64", T=14, side=14, C=64, thr=0.80, root=1, seed=7, smooth=True),
    # (no bfloat16 vectors: ATen has no CPU avg_pool3d for it, the reference only runs that dtype on a GPU)
    dict(name="oct_t12_s6_rm1", T=12, side=6, C=16, thr=0.85, root=-1, seed=8),              # root = leaf level: every leaf emitted
]


def to_np(t):
    return t.contiguous().view(torch.int16).numpy() if t.dtype == torch.bfloat16 else t.contiguous().numpy()


def main():
    for c in CASES:
        kw = dict(c=0.15, p_static=0.7) if c.get("smooth") else {}
        dtype = getattr(torch, c.get("dtype", "float32"))
        x = synth_video(c["T"], c["C"], c["side"], c["side"], seed=c["seed"], dtype=dtype, **kw)       # [T, C, H, W]
        feat = get_octree_features(x, c["thr"], c["root"])
        meta = dict(c, fn="octree", dtype=c.get("dtype", "float32"))
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), meta=json.dumps(meta),
                            x_thwc=to_np(x.permute(0, 2, 3, 1)), feat=to_np(feat))
        print(c["name"], tuple(x.shape), "->", tuple(feat.shape))


if __name__ == "__main__":
    main()
