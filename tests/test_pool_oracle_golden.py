"""The pooling oracle (oracle/pool_oracle.py) against the vectors made by the reference's get_2dPool."""
import os

import pytest

from oracle import pool_oracle as P
from tests._golden import POOL_GOLDEN as GOLDEN, load_pool_case, pool_close_enough as close_enough


def test_there_are_pool_vectors():
    assert len(GOLDEN) >= 9


@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_pool_oracle_matches_reference_vectors(path):
    meta, x, y = load_pool_case(path)
    side = meta.get("width", meta["side"])
    out = P.get_2dpool(x, meta["stride"], side, side, meta["mode"])
    assert out.shape[1] == meta["out_tokens"]
    assert close_enough(out, y, meta)


@pytest.mark.parametrize("side,tgt", [(14, 10), (14, 7), (14, 5), (10, 7), (27, 14), (7, 14), (14, 14), (13, 6)])
def test_resize_nearest_oracle_equals_torch_interpolate(side, tgt):
    """The "pyrd" baseline calls F.interpolate(video, size=(s, s)) (pyrd_attn_monkey_patch.py:100): default mode nearest."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(side * 100 + tgt)
    x = torch.randn(3, side * side, 8, generator=g)
    exp = F.interpolate(x.view(3, side, side, 8).permute(0, 3, 1, 2), size=(tgt, tgt)).permute(0, 2, 3, 1).reshape(3, tgt * tgt, 8)
    out = P.resize_nearest(x, side, side, (tgt, tgt))
    assert torch.equal(out, exp)
