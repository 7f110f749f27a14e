"""The DyCoke stage-1 oracle (oracle/dycoke_oracle.py) against the vectors made by the reference's dycoke_ttm, and the
hook glue with the oracle injected."""
import os

import pytest
import torch

from oracle import dycoke_oracle as D
from tests._golden import DYCOKE_GOLDEN, load_dycoke_case


def test_there_are_dycoke_vectors():
    assert len(DYCOKE_GOLDEN) >= 6


@pytest.mark.parametrize("path", DYCOKE_GOLDEN, ids=os.path.basename)
def test_dycoke_oracle_matches_reference_vectors(path):
    meta, x, feat, idx = load_dycoke_case(path)
    f, i = D.dycoke_ttm(x, meta["T"], meta["prune"])
    assert torch.equal(i, idx) and torch.equal(f, feat)


def test_dycoke_short_clip_raises_like_the_reference():
    for T in (1, 2, 4):
        with pytest.raises(RuntimeError):
            D.dycoke_ttm(torch.randn(T * 9, 8), T, 0.7)


def test_dycoke_hook_glue_positions():
    from sttm_amd import patch_hooks
    T, P, C, start = 6, 9, 8, 3
    g = torch.Generator().manual_seed(0)
    hs = torch.randn(1, start + T * P + 4, C, generator=g)
    pos = torch.arange(hs.shape[1]).unsqueeze(0)
    merged, p2, idx = patch_hooks.dycoke_merge(hs, pos, start, T * P, T, D.dycoke_ttm, 0.5)
    f, i = D.dycoke_ttm(hs[0, start:start + T * P], T, 0.5)
    assert torch.equal(idx, i) and torch.equal(merged[0, start:start + f.shape[0]], f)
    assert torch.equal(merged[0, :start], hs[0, :start]) and torch.equal(merged[0, start + f.shape[0]:], hs[0, start + T * P:])
    assert torch.equal(p2, pos[:, :merged.shape[1]])                       # LLaVA: truncated
    pos3 = torch.arange(3 * hs.shape[1]).reshape(3, 1, -1)
    merged, p3, idx = patch_hooks.dycoke_merge(hs, pos3, start, T * P, T, D.dycoke_ttm, 0.5, gather_positions=True)
    assert torch.equal(p3[:, :, start:start + i.shape[0]], pos3[:, :, start:start + T * P][:, :, i])     # Qwen2-VL: gathered
    assert p3.shape[-1] == merged.shape[1]
