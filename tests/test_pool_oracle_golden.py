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
