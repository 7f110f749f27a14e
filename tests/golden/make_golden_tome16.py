"""Golden vectors of the ToMe baseline on 16-bit inputs (bfloat16 / float16), produced by the REFERENCE's own
`get_tome_features` on CPU (token_merging_utils/tome_interface.py:3-9 -> tome_token_merger.py:133-152).

    python tests/golden/make_golden_tome16.py          # in the build container (needs /root/reference)

The reference hands `get_tome_features` the decoder's hidden states (tome_attn_monkey_patch.py:88-107), which are bf16 in
production, so every intermediate of tome_per_video is rounded to that dtype.  Two things in it are NOT reproducible
bit for bit by any implementation: `argsort(descending=True)` is not stable (bf16 scores tie massively -- which of the tied
tokens make the top-r cut is the sort's business), and the fp32 accumulation order of the bf16 matmul.  The GPU tests
therefore compare kept-token ids as sets (>= 97 % agreement) and features on the ids both sides kept.
Only inputs and outputs are stored: no reference code.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from sttm_amd.synth import synth_video                                   # noqa: E402
from token_merging_utils.tome_interface import get_tome_features       # noqa: E402  (reference)

CASES = [
    dict(name="tome16_bf16_050", T=4, C=64, H=14, W=14, seed=130, ratio=0.50, n_head=1, dtype="bfloat16"),
    dict(name="tome16_bf16_070", T=4, C=64, H=14, W=14, seed=131, ratio=0.70, n_head=1, dtype="bfloat16"),
    dict(name="tome16_bf16_085", T=6, C=128, H=14, W=14, seed=132, ratio=0.85, n_head=1, dtype="bfloat16"),
    dict(name="tome16_bf16_070_h4", T=3, C=128, H=14, W=14, seed=133, ratio=0.70, n_head=4, dtype="bfloat16"),
    dict(name="tome16_bf16_050_odd", T=3, C=64, H=7, W=7, seed=134, ratio=0.50, n_head=1, dtype="bfloat16"),
    dict(name="tome16_f16_050", T=4, C=64, H=14, W=14, seed=135, ratio=0.50, n_head=1, dtype="float16"),
    dict(name="tome16_f16_070", T=4, C=64, H=14, W=14, seed=136, ratio=0.70, n_head=1, dtype="float16"),
    dict(name="tome16_f16_085", T=6, C=128, H=14, W=14, seed=137, ratio=0.85, n_head=1, dtype="float16"),
]


def to_np(t):
    t = t.contiguous()
    return t.view(torch.int16).numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.numpy()


def main():
    for case in CASES:
        dt = getattr(torch, case["dtype"])
        x = synth_video(case["T"], case["C"], case["H"], case["W"], seed=case["seed"], dtype=dt)
        feat, idx = get_tome_features(x, case["ratio"], "video", case["n_head"])
        assert feat.dtype == dt and idx.dtype == torch.int64
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"),
                            x_thwc=to_np(x.permute(0, 2, 3, 1)), feat=to_np(feat), idx=to_np(idx),
                            meta=json.dumps(dict(case, fn="tome", torch=torch.__version__,
                                                 # ATen's bf16/fp16 CPU scatter_add rounds once (expanded-index path) only in builds with
                                                 # FBGEMM + OpenMP; other builds round per add (csrc/tome.hip, k_tome_merge)
                                                 torch_parallel=torch.__config__.parallel_info().splitlines()[:4],
                                                 torch_has_fbgemm="USE_FBGEMM=ON" in torch.__config__.show())))
        print(case["name"], tuple(feat.shape))


if __name__ == "__main__":
    main()
