"""GPU parity tests of the octree baseline (sttm_amd.octree_utils.get_octree_features -> sttm_octree_build)."""
import os

import pytest
import torch

from tests._golden import OCTREE_GOLDEN, load_octree_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("path", OCTREE_GOLDEN, ids=os.path.basename)
def test_octree_golden_vectors(path):
    from sttm_amd.octree_utils import get_octree_features
    meta, x, feat = load_octree_case(path)
    out = get_octree_features(x.to(DEV), meta["thr"], meta["root"]).cpu()
    assert out.shape == feat.shape, f"{out.shape} vs {feat.shape}"
    assert float((out.float() - feat.float()).abs().max()) <= 1e-5


@pytest.mark.parametrize("T,C,side,dtype,thr,root,kind", [
    (30, 256, 14, torch.float32, 0.85, 0, "synth"), (28, 1024, 14, torch.float32, 0.80, 1, "smooth"),
    (29, 128, 14, torch.bfloat16, 0.85, 1, "smooth"), (16, 64, 8, torch.float16, 0.8, 0, "synth"),
    (27, 100, 27, torch.float32, 0.85, 1, "smooth"), (21, 34, 7, torch.float32, 0.7, -2, "iid"),
    (128, 1024, 14, torch.float32, 0.85, 1, "synth"),                                 # the headline clip: 9 cubes + 2 frames
])
def test_octree_against_oracle(T, C, side, dtype, thr, root, kind):
    from oracle import octree_oracle as OC
    from sttm_amd.octree_utils import get_octree_features
    from sttm_amd.synth import iid_video, synth_video
    if kind == "iid":
        x = iid_video(T, C, side, side, seed=T, dtype=dtype)
    else:
        x = synth_video(T, C, side, side, seed=T, dtype=dtype, **(dict(c=0.15, p_static=0.7) if kind == "smooth" else {}))
    exp = OC.get_octree_features(x, thr, root)
    out = get_octree_features(x.to(DEV), thr, root).cpu()
    assert out.shape == exp.shape, f"{out.shape} vs {exp.shape}"
    tol = 1e-5 if dtype == torch.float32 else 0.0
    if dtype == torch.float32:
        assert float((out - exp).abs().max()) <= tol
    else:
        assert torch.equal(out, exp)                              # the pooling arithmetic is identical, one rounding per level


def test_octree_interface_behaviour():
    from sttm_amd.octree_utils import get_octree_features
    from sttm_amd.synth import synth_video
    x = synth_video(14, 16, 14, 14, seed=0).to(DEV)
    with pytest.raises(IndexError):
        get_octree_features(x, 0.85, 9)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_octree_features(x.cpu(), 0.85, 0)
    # threshold above every cosine: every leaf of the cube is emitted, in order
    out = get_octree_features(x, 2.0, 0)
    assert torch.equal(out, x.permute(0, 2, 3, 1).reshape(-1, 16))
    # threshold below every cosine: only the root cells remain
    out = get_octree_features(x, -2.0, 0)
    assert out.shape[0] == 8


def test_octree_pattern_runs_on_device():
    transformers = pytest.importorskip("transformers")
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    from oracle import octree_oracle as OC
    from sttm_amd import monkey_patch_interface as MPI
    from sttm_amd import patch_hooks
    from sttm_amd.synth import synth_video
    torch.manual_seed(0)
    C, T, side, start = 64, 9, 7, 4
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval().to(DEV)
    vis = synth_video(T, C, side, side, seed=3, c=0.15, p_static=0.7).permute(0, 2, 3, 1).reshape(1, T * side * side, C)
    hs = torch.cat([torch.randn(1, start, C), vis, torch.randn(1, 6, C)], 1).to(DEV)
    try:
        MPI.replace_qwen2_by_sparse_attn("octree", sa_start_layer_idx=0, sa_tree_thresh=0.8, sa_tree_root_level=0)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(T * side * side)
        model.num_frame = torch.tensor(T)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
            pos = torch.arange(hs.shape[1]).unsqueeze(0)
            hm, pm = patch_hooks.octree_merge(hs.cpu(), pos, start, T * side * side, T, OC.get_octree_features, 0.8, 0)
            hm, pm = hm.to(DEV), pm.to(DEV)
            pe = model.rotary_emb(hm, pm)
            for layer in model.layers:
                hm = layer(hm, attention_mask=None, position_embeddings=pe, position_ids=pm)
            ref = model.norm(hm)
        assert out.shape == ref.shape and out.shape[1] < hs.shape[1]
        assert torch.allclose(out, ref, atol=2e-5)
    finally:
        MPI.restore_qwen2()
