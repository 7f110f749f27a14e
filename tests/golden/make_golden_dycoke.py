#!/usr/bin/env python3
"""Golden vectors for the DyCoke stage-1 baseline, made by RUNNING THE REFERENCE's `dycoke_ttm`
(token_merging_utils/dycoke_merger.py:8-83) in the build container (CPU).  Only data is written.

    python tests/golden/make_golden_dycoke.py        # rewrites tests/golden/dyc_*.npz
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

from sttm_amd.synth import synth_video                                  # noqa: E402
from token_merging_utils.dycoke_merger import dycoke_ttm               # noqa: E402  (reference)

CASES = [
    dict(name="dyc_t8_p196", T=8, side=14, C=64, prune=0.7, seed=1),
    dict(name="dyc_t9_odd", T=9, side=14, C=32, prune=0.7, seed=2),
    dict(name="dyc_t5_min", T=5, side=7, C=16, prune=0.5, seed=3),
    dict(name="dyc_t12_p100_r03", T=12, side=10, C=48, prune=0.3, seed=4),
    dict(name="dyc_t16_c256", T=16, side=14, C=256, prune=0.85, seed=5),
    dict(name="dyc_t6_p9_keep0", T=6, side=3, C=8, prune=0.95, seed=6),       # k = int(0.05 * 9) = 0
]


def main():
    for c in CASES:
        vid = synth_video(c["T"], c["C"], c["side"], c["side"], seed=c["seed"])               # [T, C, H, W]
        g = torch.Generator().manual_seed(100 + c["seed"])
        x = vid.permute(0, 2, 3, 1).reshape(c["T"] * c["side"] ** 2, c["C"]).contiguous()
        x = x + 1e-3 * torch.randn(x.shape, generator=g)                                     # no exactly repeated tokens -> no topk ties
        feat, idx = dycoke_ttm(x, c["T"], c["prune"])
        meta = dict(c, fn="dycoke")
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), meta=json.dumps(meta), x=x.numpy(), feat=feat.numpy(),
                            idx=idx.numpy())
        print(c["name"], tuple(x.shape), "->", tuple(feat.shape))


if __name__ == "__main__":
    main()
