"""Pin the CPU oracle to the reference: it must reproduce every golden vector bit-for-bit."""
import os

import pytest
import torch

from oracle import sttm_oracle as O
from tests._golden import case_paths, kat, load_case, quadtree_kwargs


@pytest.mark.parametrize("path", case_paths(["sp_", "st_", "sl_", "pe_"]), ids=os.path.basename)
def test_quadtree_matches_reference(path):
    c = load_case(path)
    thr, kw = quadtree_kwargs(c["meta"])
    if "pos_embs" in c:
        feat, npatch, tlbr, pos = O.get_quadtree_features(c["x"], thr, pos_embs=c["pos_embs"], **kw)
        for got, exp in zip(pos, c["out_pos"]):
            assert got.dtype == exp.dtype and torch.equal(got.float(), exp.float())
    else:
        feat, npatch, tlbr = O.get_quadtree_features(c["x"], thr, **kw)
    assert tlbr.dtype == torch.int32 and npatch.dtype == torch.int32
    assert feat.dtype == c["feat"].dtype
    assert torch.equal(tlbr, c["tlbr"])
    assert torch.equal(npatch, c["npatch"])
    assert torch.equal(feat.view(torch.int16 if feat.dtype == torch.bfloat16 else torch.int32),
                       c["feat"].view(torch.int16 if feat.dtype == torch.bfloat16 else torch.int32))


@pytest.mark.parametrize("path", case_paths(["tome_"]), ids=os.path.basename)
def test_tome_matches_reference(path):
    c = load_case(path)
    m = c["meta"]
    feat, idx = O.get_tome_features(c["x"], m["ratio"], "video", m["n_head"])
    assert idx.dtype == torch.int64
    assert torch.equal(idx, c["idx"])
    assert torch.equal(feat.view(torch.int32), c["feat"].view(torch.int32))


@pytest.mark.parametrize("case", kat()["label"], ids=lambda c: c["name"])
def test_label_propagation_kat(case):
    pairs = torch.tensor(case["pairs"], dtype=torch.int64).reshape(-1, 2)
    rep, _ = O.propagate_labels(pairs, case["N"])
    assert rep.tolist() == case["rep"]


def test_label_propagation_is_not_connected_components():
    # the iteration stops on idempotency, leaving 1 and 3 apart from 0 and 2
    rep, _ = O.propagate_labels(torch.tensor([[0, 2], [1, 3], [2, 3]]), 4)
    assert rep.tolist() == [0, 1, 0, 1]


@pytest.mark.parametrize("case", kat()["errors"], ids=lambda c: c["name"])
def test_error_behaviour(case):
    from sttm_amd.synth import synth_video
    x = synth_video(case["T"], case["C"], case["H"], case["W"], seed=99)

    def run():
        if case["fn"] == "quadtree":
            kw = dict(case["kw"])
            if case.get("pos"):
                pe = torch.rand(case["T"], case["H"], case["W"], case["pos"]).permute(0, 3, 1, 2)
                kw["pos_embs"] = (pe, pe.clone())
            return O.get_quadtree_features(x, kw.pop("threshold"), **kw)
        return O.get_tome_features(x, **case["kw"])

    if case["raises"]:
        with pytest.raises(getattr(__import__("builtins"), case["raises"])):
            run()
    else:
        out = run()
        assert (out is None) == case["returns_none"]
        if "n_out" in case:
            assert out[0].shape[0] == case["n_out"]


@pytest.mark.parametrize("path", case_paths(["sp_14_r1_085", "sp_27_r1_085", "sp_13x24", "sp_20x36", "sp_10x30_r0"]),
                         ids=os.path.basename)
def test_candidate_pairs_two_ways(path):
    """The dense box test (the reference's method) equals the leaf-owner construction."""
    c = load_case(path)
    tlbr = c["tlbr"]
    H, W = c["meta"]["H"], c["meta"]["W"]
    a = O.candidate_pairs(tlbr)
    b = O.candidate_pairs_by_owner(tlbr, H, W)
    assert torch.equal(torch.unique(a, dim=0), b)


def test_level_sizes_quirk_q5():
    assert O.level_sizes(10, 30) == [(2, 4), (3, 8), (5, 15), (10, 30)]
    assert O.level_sizes(14, 14) == [(2, 2), (4, 4), (7, 7), (14, 14)]
    assert O.level_sizes(20, 36) == [(2, 3), (3, 5), (5, 9), (10, 18), (20, 36)]


@pytest.mark.parametrize("path", case_paths(["tome16_"]), ids=os.path.basename)
def test_oracle_tome_16bit_vectors(path):
    """ToMe on bfloat16 / float16 inputs: the oracle issues the same ATen calls as the reference on the input dtype, so on the
    machine that made the vectors it reproduces them exactly (ids and features)."""
    from oracle import sttm_oracle as O
    c = load_case(path)
    m = c["meta"]
    feat, idx = O.get_tome_features(c["x"], m["ratio"], "video", m["n_head"])
    assert feat.dtype == c["feat"].dtype and torch.equal(idx, c["idx"])
    assert torch.equal(feat.view(torch.int16), c["feat"].view(torch.int16))
