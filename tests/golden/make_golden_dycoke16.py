"""Golden vectors of DyCoke stage-1 pruning on 16-bit inputs (bfloat16 / float16), produced by the REFERENCE's own
`dycoke_ttm` on CPU (token_merging_utils/dycoke_merger.py:8-83).

    python tests/golden/make_golden_dycoke16.py         # in the build container (needs /root/reference)

On 16-bit inputs F.cosine_similarity rounds every intermediate to the input dtype, so a frame's similarities take few
distinct values and `topk` breaks the ties in an unspecified order: the GPU tests compare the kept tokens of every frame as
SETS, up to the tokens tied with the cut value.  The per-token similarities are stored too (the comparison needs them).
Only inputs and outputs are stored: no reference code.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.environ.get("STTM_REFERENCE", "/root/reference"))

from sttm_amd.synth import synth_video                                  # noqa: E402
from token_merging_utils.dycoke_merger import dycoke_ttm               # noqa: E402  (reference)

CASES = [
    dict(name="dyc16_bf16_t8", T=8, side=14, C=64, prune=0.7, seed=11, dtype="bfloat16"),
    dict(name="dyc16_bf16_t9_c256", T=9, side=14, C=256, prune=0.5, seed=12, dtype="bfloat16"),
    dict(name="dyc16_f16_t8", T=8, side=14, C=64, prune=0.7, seed=13, dtype="float16"),
    dict(name="dyc16_f16_t12_c100", T=12, side=10, C=100, prune=0.3, seed=14, dtype="float16"),
]


def main():
    for c in CASES:
        dt = getattr(torch, c["dtype"])
        vid = synth_video(c["T"], c["C"], c["side"], c["side"], seed=c["seed"], dtype=dt)
        x = vid.permute(0, 2, 3, 1).reshape(c["T"] * c["side"] ** 2, c["C"]).contiguous()
        feat, idx = dycoke_ttm(x, c["T"], c["prune"])
        P = c["side"] ** 2
        fr = x.reshape(c["T"], P, c["C"])
        # similarity of every pruned frame to its partner, as the reference computes it (ATen on the input dtype)
        sims = np.zeros((c["T"], P), dtype=np.float32)
        for f in range(c["T"]):
            partner = f - 1 if f % 2 == 1 else (f - 2 if (f % 4 == 2 and f - 2 < c["T"] - 4) else None)
            if partner is not None:
                sims[f] = F.cosine_similarity(fr[partner], fr[f], dim=1).float().numpy()
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), meta=json.dumps(dict(c, fn="dycoke")),
                            x=x.view(torch.int16).numpy(), feat=feat.contiguous().view(torch.int16).numpy(), idx=idx.numpy(), sims=sims)
        print(c["name"], tuple(x.shape), "->", tuple(feat.shape))


if __name__ == "__main__":
    main()
