"""Fake-decoder tests of the L2 glue: slicing / concat / position ids around the merge, and the
installer's names + class-attribute mechanism.  The merge function is injected (the CPU oracle), so this
runs without a GPU; the HIP-backed default is exercised by the gpu-marked test at the bottom."""
import math

import pytest
import torch

from oracle import sttm_oracle as O
from sttm_amd import monkey_patch_interface as MPI
from sttm_amd import patch_hooks
from sttm_amd.synth import synth_video


def _prompt(T=4, H=14, W=14, C=32, n_sys=5, n_inst=7, seed=0):
    vis = synth_video(T, C, H, W, seed=seed).permute(0, 2, 3, 1).reshape(1, T * H * W, C)
    g = torch.Generator().manual_seed(seed + 1)
    hs = torch.cat([torch.randn(1, n_sys, C, generator=g), vis, torch.randn(1, n_inst, C, generator=g)], dim=1)
    return hs, n_sys, T * H * W


def test_llava_hook_slices_merges_and_truncates_positions():
    hs, start, length = _prompt()
    pos = torch.arange(hs.shape[1]).unsqueeze(0)
    out, pos2, idx = patch_hooks.quadtree_merge_llava(hs, pos, start, length, 4, O.get_quadtree_features, 0.85, 0.55, 1, False)
    video = hs[0, start:start + length].reshape(4, 14, 14, -1).permute(0, 3, 1, 2)
    f, _, t = O.get_quadtree_features(video, 0.85, 0.55, 1)
    assert torch.equal(out[:, :start], hs[:, :start]) and torch.equal(out[:, start + f.shape[0]:], hs[:, start + length:])
    assert torch.equal(out[0, start:start + f.shape[0]], f)
    assert torch.equal(pos2, pos[:, :out.shape[1]])
    assert torch.equal(idx, t[:, 0] * 196 + t[:, 1] * 14 + t[:, 2])
    assert not out.data_ptr() == hs.data_ptr()


def test_qwen2vl_hook_gathers_3d_positions():
    T, H, W = 3, 10, 18
    hs, start, length = _prompt(T, H, W, C=16)
    S = hs.shape[1]
    pos = torch.stack([torch.arange(S), torch.arange(S) * 2, torch.arange(S) * 3]).unsqueeze(1)      # [3, 1, S]
    out, pos2, cache_pos, idx = patch_hooks.quadtree_merge_qwen2vl(hs, pos, start, length, T, H, W,
                                                                   O.get_quadtree_features, 0.85, 0.6, 1, False)
    n = idx.shape[0]
    assert out.shape[1] == S - length + n and pos2.shape == (3, 1, out.shape[1])
    assert torch.equal(pos2[:, :, :start], pos[:, :, :start])
    assert torch.equal(pos2[:, :, start:start + n], pos[:, :, start:start + length][:, :, idx.long()])
    assert torch.equal(pos2[:, :, start + n:], pos[:, :, start + length:])
    assert torch.equal(cache_pos, torch.arange(out.shape[1], dtype=torch.int))


def test_tome_hook():
    hs, start, length = _prompt(T=3, C=16)
    pos = torch.arange(hs.shape[1]).unsqueeze(0)
    out, pos2, tok = patch_hooks.tome_merge(hs, pos, start, length, 3, O.get_tome_features, 0.5, "video")
    assert out.shape[1] == hs.shape[1] - length + math.ceil(length * 0.5)
    assert torch.equal(pos2, pos[:, :out.shape[1]])
    assert tok.dtype == torch.int64


def test_qwen2vl_tome_hook_gathers_3d_positions_by_token_index():
    """token_merging_qwen2vl_monkey_patch/tome_attn_monkey_patch.py:105-108: system ids ++ visual ids GATHERED by the ToMe
    token index ++ instruction ids (not a truncation).  Expected values are written out here from those lines, independently
    of patch_hooks."""
    T, H, W = 3, 10, 18
    hs, start, length = _prompt(T, H, W, C=16)
    S, end = hs.shape[1], start + length
    pos = torch.stack([torch.arange(S), 1000 + torch.arange(S) * 2, 5000 + torch.arange(S) * 3]).unsqueeze(1)      # [3, 1, S]
    for ratio in (0.5, 0.7):
        out, pos2, tok = patch_hooks.tome_merge(hs, pos, start, length, T, O.get_tome_features, ratio, "video", H=H, W=W,
                                                gather_positions=True)
        video = hs[0, start:end].reshape(T, H, W, -1).permute(0, 3, 1, 2)
        ef, ei = O.get_tome_features(video, ratio, "video")
        n = ei.shape[0]
        assert torch.equal(tok, ei) and torch.equal(out[0, start:start + n], ef)
        assert pos2.shape == (3, 1, S - length + n)
        expect = torch.empty(3, 1, S - length + n, dtype=pos.dtype)
        for a in range(3):
            for k in range(start):
                expect[a, 0, k] = pos[a, 0, k]
            for k in range(n):
                expect[a, 0, start + k] = pos[a, 0, start + int(ei[k])]
            for k in range(S - end):
                expect[a, 0, start + n + k] = pos[a, 0, end + k]
        assert torch.equal(pos2, expect)
        # the instruction tokens keep their ORIGINAL ids: a truncation would have given them the visual ids
        assert torch.equal(pos2[:, :, start + n:], pos[:, :, end:]) and not torch.equal(pos2, pos[:, :, :pos2.shape[2]])


def test_patched_forward_stays_causal_with_eager_attention():
    """After the merge the masks are rebuilt for the shorter sequence: with eager attention (where a None mask would mean
    bidirectional attention) the patched forward equals the manual forward with an explicit causal mask."""
    pytest.importorskip("transformers")
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    torch.manual_seed(0)
    C = 32
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="eager")
    model = Qwen2Model(cfg).eval()
    hs, start, length = _prompt(T=4, C=C)

    def causal(n):
        m = torch.full((n, n), float("-inf")).triu(1)
        return m[None, None]
    try:
        MPI.replace_qwen2_by_sparse_attn("quadtree", sa_start_layer_idx=1, sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.55,
                                         sa_tree_root_level=1)
        Qwen2Model.sttm_merge_fn = staticmethod(O.get_quadtree_features)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(4)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=True).last_hidden_state        # with a cache: earlier layers hold the LONG keys
            pos = torch.arange(hs.shape[1]).unsqueeze(0)
            pe = model.rotary_emb(hs, pos)
            h = model.layers[0](hs, attention_mask=causal(hs.shape[1]), position_embeddings=pe, position_ids=pos)
            h, pos, _ = patch_hooks.quadtree_merge_llava(h, pos, start, length, 4, O.get_quadtree_features, 0.85, 0.55, 1, False)
            pe = model.rotary_emb(h, pos)
            for layer in model.layers[1:]:
                h = layer(h, attention_mask=causal(h.shape[1]), position_embeddings=pe, position_ids=pos)
            ref = model.norm(h)
        assert out.shape == ref.shape and out.shape[1] < hs.shape[1]
        assert torch.allclose(out, ref, atol=1e-5)
    finally:
        MPI.restore_qwen2()
        if "sttm_merge_fn" in Qwen2Model.__dict__:
            del Qwen2Model.sttm_merge_fn


def test_installer_names_and_errors():
    for name in ("quadtree_vis", "dycoke", "nonsense"):
        with pytest.raises(NotImplementedError):
            MPI.replace_qwen2_by_sparse_attn(name)


def test_patched_qwen2_forward_matches_manual_layers():
    """Install the patch on a tiny random Qwen2Model and compare with running the layers by hand."""
    transformers = pytest.importorskip("transformers")
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    torch.manual_seed(0)
    C = 32
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval()
    hs, start, length = _prompt(T=4, C=C)
    try:
        MPI.replace_qwen2_by_sparse_attn("quadtree", sa_start_layer_idx=1, sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.55,
                                         sa_tree_root_level=1, unused_flag=123)
        Qwen2Model.sttm_merge_fn = staticmethod(O.get_quadtree_features)          # inject the CPU oracle
        assert Qwen2Model.sa_tree_thresh == 0.85 and Qwen2Model.sa_start_layer_idx == 1     # class attributes
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(4)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
            # by hand
            pos = torch.arange(hs.shape[1]).unsqueeze(0)
            h = hs
            pe = model.rotary_emb(h, pos)
            h = model.layers[0](h, attention_mask=None, position_embeddings=pe, position_ids=pos)
            h, pos, _ = patch_hooks.quadtree_merge_llava(h, pos, start, length, 4, O.get_quadtree_features, 0.85, 0.55, 1, False)
            pe = model.rotary_emb(h, pos)
            for layer in model.layers[1:]:
                h = layer(h, attention_mask=None, position_embeddings=pe, position_ids=pos)
            ref = model.norm(h)
        assert out.shape == ref.shape and out.shape[1] < hs.shape[1]
        assert torch.allclose(out, ref, atol=1e-5)
    finally:
        MPI.restore_qwen2()
        if hasattr(Qwen2Model, "sttm_merge_fn"):
            del Qwen2Model.sttm_merge_fn


def test_patched_qwen2vl_forward_matches_manual_layers():
    """Qwen2-VL text model: 3-D mRoPE position ids are gathered by the merged-token index."""
    pytest.importorskip("transformers")
    try:
        from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLTextConfig
        from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextModel
    except Exception:  # noqa: BLE001
        pytest.skip("this transformers has no Qwen2VLTextModel")
    torch.manual_seed(0)
    C, T, H, W = 32, 3, 10, 18
    cfg = Qwen2VLTextConfig(vocab_size=64, hidden_size=C, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                            num_key_value_heads=2, max_position_embeddings=4096,
                            rope_parameters={"rope_type": "default", "mrope_section": [1, 1, 2], "rope_theta": 10000.0})
    cfg._attn_implementation = "sdpa"
    model = Qwen2VLTextModel(cfg).eval()
    hs, start, length = _prompt(T, H, W, C=C)
    S = hs.shape[1]
    pos = torch.stack([torch.arange(S), torch.arange(S) // 2, torch.arange(S) // 3]).unsqueeze(1)     # [3, 1, S]
    try:
        MPI.replace_qwen2_by_sparse_attn("quadtree", sa_start_layer_idx=1, sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.6,
                                         sa_tree_root_level=1)
        Qwen2VLTextModel.sttm_merge_fn = staticmethod(O.get_quadtree_features)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(T)
        model.image_H = torch.tensor(H)
        model.image_W = torch.tensor(W)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, position_ids=pos, use_cache=False).last_hidden_state
            h = hs
            pe = model.rotary_emb(h, pos)
            h = model.layers[0](h, attention_mask=None, position_embeddings=pe, position_ids=None)
            h, p2, _, _ = patch_hooks.quadtree_merge_qwen2vl(h, pos, start, length, T, H, W, O.get_quadtree_features, 0.85, 0.6, 1, False)
            pe = model.rotary_emb(h, p2)
            for layer in model.layers[1:]:
                h = layer(h, attention_mask=None, position_embeddings=pe, position_ids=None)
            ref = model.norm(h)
        assert out.shape == ref.shape and out.shape[1] < S
        assert torch.allclose(out, ref, atol=1e-5)
    finally:
        MPI.restore_qwen2()
        for cls in (Qwen2VLTextModel,):
            if "sttm_merge_fn" in cls.__dict__:
                del cls.sttm_merge_fn
        from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
        if "sttm_merge_fn" in Qwen2Model.__dict__:
            del Qwen2Model.sttm_merge_fn


@pytest.mark.gpu
def test_hook_on_gpu_uses_the_hip_path():
    from sttm_amd import get_quadtree_features
    dev = torch.device("cuda:0")
    hs, start, length = _prompt(T=6, C=256)
    pos = torch.arange(hs.shape[1]).unsqueeze(0)
    out, pos2, idx = patch_hooks.quadtree_merge_llava(hs.to(dev), pos.to(dev), start, length, 6, get_quadtree_features,
                                                      0.85, 0.55, 1, False)
    ref, _, ridx = patch_hooks.quadtree_merge_llava(hs, pos, start, length, 6, O.get_quadtree_features, 0.85, 0.55, 1, False)
    assert torch.equal(idx.cpu(), ridx) and float((out.cpu() - ref).abs().max()) <= 1e-5


def test_pyrd_pattern_forward_matches_manual_layers():
    """"pyrd" baseline (pyrd_attn_monkey_patch.py:88-112, installer :167-173): frames resized (nearest) before the listed
    layers, possibly more than once; checked against the manual forward with torch's own F.interpolate."""
    transformers = pytest.importorskip("transformers")
    import torch.nn.functional as F
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    from oracle import pool_oracle as PO
    torch.manual_seed(0)
    C, T = 32, 4
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=64, num_hidden_layers=4, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval()
    hs, start, length = _prompt(T=T, C=C)
    try:
        Qwen2Model.sttm_resize_fn = staticmethod(PO.resize_nearest)             # inject the CPU oracle
        MPI.replace_qwen2_by_sparse_attn("pyrd", sa_pyrd_loc_list=[1, 3], sa_pyrd_size_list=[10, 6])
        assert Qwen2Model.sa_pyrd_idx2size == {1: 10, 3: 6}
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(T)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
            assert int(model.image_token_length) == T * 36                       # persists, like the reference (:101)
            pos = torch.arange(hs.shape[1]).unsqueeze(0)
            h, cur = hs, length
            for i, layer in enumerate(model.layers):
                if i in (1, 3):
                    tgt = {1: 10, 3: 6}[i]
                    side = int((cur // T) ** 0.5)
                    vis = h[0, start:start + cur].reshape(T, side, side, C).permute(0, 3, 1, 2)
                    r = F.interpolate(vis, size=(tgt, tgt)).permute(0, 2, 3, 1).reshape(1, T * tgt * tgt, C)
                    h = torch.cat([h[:, :start], r, h[:, start + cur:]], dim=1)
                    cur = T * tgt * tgt
                    pos = pos[:, :h.shape[1]]
                pe = model.rotary_emb(h, pos)
                h = layer(h, attention_mask=None, position_embeddings=pe, position_ids=pos)
            ref = model.norm(h)
        assert out.shape == ref.shape and out.shape[1] == hs.shape[1] - length + T * 36
        assert torch.allclose(out, ref, atol=1e-5)
    finally:
        MPI.restore_qwen2()
        for name in ("sttm_resize_fn",):
            if name in Qwen2Model.__dict__:
                delattr(Qwen2Model, name)



@pytest.mark.parametrize("ver", [0, 1, 2])
def test_abl_pos_pattern_forward_matches_manual_layers(ver):
    """Position-embedding ablation (quadtree_attn_monkey_patch_for_abl_pos.py:88-136) with the oracle injected."""
    transformers = pytest.importorskip("transformers")
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    torch.manual_seed(0)
    C, T = 32, 4
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval()
    hs, start, length = _prompt(T=T, C=C)
    try:
        MPI.replace_qwen2_by_sparse_attn("quadtree-abl-pos", sa_start_layer_idx=1, sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.55,
                                         sa_tree_root_level=1, pos_emb_ver=ver, pos_emb_weighted_avg=False)
        Qwen2Model.sttm_merge_fn = staticmethod(O.get_quadtree_features)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(T)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
            pos = torch.arange(hs.shape[1]).unsqueeze(0)
            pe = model.rotary_emb(hs, pos)
            h = model.layers[0](hs, attention_mask=None, position_embeddings=pe, position_ids=pos)
            H = int(math.sqrt(length // T))
            end = start + length
            video = h[0, start:end].reshape(T, H, H, C).permute(0, 3, 1, 2)
            if ver == 1:
                pv = tuple(p[0, start:end].reshape(T, H, H, -1).permute(0, 3, 1, 2) for p in pe)
                f, _, tl, mp = O.get_quadtree_features(video, 0.85, 0.55, 1, False, pos_embs=pv, pos_emb_weighted_avg=False)
            else:
                f, _, tl = O.get_quadtree_features(video, 0.85, 0.55, 1, False)
            idx = (tl[:, 0] * H * H + tl[:, 1] * H + tl[:, 2]).long()
            h = torch.cat([h[:, :start], f.unsqueeze(0), h[:, end:]], 1)
            if ver == 0:
                pos = pos[:, :h.shape[1]]; pe = model.rotary_emb(h, pos)
            elif ver == 1:
                pe = tuple(torch.cat([p[:, :start], m.unsqueeze(0), p[:, end:]], 1) for p, m in zip(pe, mp)); pos = pos[:, :h.shape[1]]
            else:
                pos = torch.cat([pos[:, :start], pos[:, start:end][:, idx], pos[:, end:]], -1); pe = model.rotary_emb(h, pos)
            for layer in model.layers[1:]:
                h = layer(h, attention_mask=None, position_embeddings=pe, position_ids=pos)
            ref = model.norm(h)
        assert out.shape == ref.shape and out.shape[1] < hs.shape[1]
        assert torch.allclose(out, ref, atol=1e-5)
    finally:
        MPI.restore_qwen2()
        if "sttm_merge_fn" in Qwen2Model.__dict__:
            del Qwen2Model.sttm_merge_fn
