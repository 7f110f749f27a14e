"""The octree oracle (oracle/octree_oracle.py) against the vectors made by the reference's get_octree_features."""
import os

import pytest
import torch

from oracle import octree_oracle as OC
from tests._golden import OCTREE_GOLDEN, load_octree_case


def test_there_are_octree_vectors():
    assert len(OCTREE_GOLDEN) >= 8


@pytest.mark.parametrize("path", OCTREE_GOLDEN, ids=os.path.basename)
def test_octree_oracle_matches_reference_vectors(path):
    meta, x, feat = load_octree_case(path)
    out = OC.get_octree_features(x, meta["thr"], meta["root"])
    assert out.shape == feat.shape
    assert float((out.float() - feat.float()).abs().max()) <= 1e-6


def test_octree_root_level_out_of_range_raises_index_error():
    from sttm_amd.synth import synth_video
    with pytest.raises(IndexError):
        OC.get_octree_features(synth_video(14, 8, 14, 14, seed=0), 0.85, 7)
