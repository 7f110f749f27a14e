"""GPU tests of the decoder-hook glue on the device path: the fused slice -> merge -> concat must give
exactly what the reference's three steps give (quadtree_attn_monkey_patch.py:88-117, qwen2vl :88-115), and the patched
Qwen2 forward must run end to end on the GPU through the HIP library."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _prompt(T, C, H, W, dtype, n_sys=7, n_inst=11, seed=0):
    from sttm_amd.synth import synth_video
    dev = torch.device("cuda:0")
    vid = synth_video(T, C, H, W, seed=seed, dtype=dtype, device=dev)                     # [T, C, H, W]
    vis = vid.permute(0, 2, 3, 1).reshape(1, T * H * W, C)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    sys_f = torch.randn(1, n_sys, C, generator=g).to(dev, dtype)
    inst_f = torch.randn(1, n_inst, C, generator=g).to(dev, dtype)
    return torch.cat([sys_f, vis, inst_f], dim=1).contiguous(), n_sys, T * H * W


@pytest.mark.parametrize("T,C,H,W,dtype", [(8, 256, 14, 14, torch.float32), (6, 512, 14, 14, torch.bfloat16),
                                           (4, 128, 27, 27, torch.float32)])
def test_fused_concat_equals_three_step_hook_llava(T, C, H, W, dtype):
    from sttm_amd import get_quadtree_features, get_quadtree_features_into, patch_hooks
    hs, start, length = _prompt(T, C, H, W, dtype)
    pos = torch.arange(hs.shape[1], device=hs.device).unsqueeze(0)
    keep = hs.clone()
    a = patch_hooks.quadtree_merge_llava(hs, pos, start, length, T, get_quadtree_features, 0.85, 0.55, 1, False)
    b = patch_hooks.quadtree_merge_llava(hs, pos, start, length, T, get_quadtree_features, 0.85, 0.55, 1, False,
                                         merge_into_fn=get_quadtree_features_into)
    assert torch.equal(hs, keep)                                   # the input hidden states are not modified
    assert b[0].shape == a[0].shape and b[0].shape[1] < hs.shape[1]
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("T,C,H,W,dtype", [(8, 256, 14, 14, torch.float32), (4, 128, 18, 26, torch.float32),
                                           (5, 256, 20, 36, torch.bfloat16)])
def test_fused_concat_equals_three_step_hook_qwen2vl(T, C, H, W, dtype):
    from sttm_amd import get_quadtree_features, get_quadtree_features_into, patch_hooks
    hs, start, length = _prompt(T, C, H, W, dtype, seed=3)
    S = hs.shape[1]
    pos = torch.arange(3 * S, device=hs.device).reshape(3, 1, S)
    a = patch_hooks.quadtree_merge_qwen2vl(hs, pos, start, length, T, H, W, get_quadtree_features, 0.85, 0.60, 0, False)
    b = patch_hooks.quadtree_merge_qwen2vl(hs, pos, start, length, T, H, W, get_quadtree_features, 0.85, 0.60, 0, False,
                                           merge_into_fn=get_quadtree_features_into)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_into_rejects_short_destination_and_leaves_tail_untouched():
    from sttm_amd import get_quadtree_features, get_quadtree_features_into
    from sttm_amd.synth import synth_video
    dev = torch.device("cuda:0")
    x = synth_video(6, 128, 14, 14, seed=5, device=dev)
    N = 6 * 196
    with pytest.raises(ValueError):
        get_quadtree_features_into(torch.empty((N - 1, 128), device=dev), x, 0.85, 0.55, 1)
    dest = torch.full((N + 5, 128), 7.0, device=dev)
    f, n, t = get_quadtree_features_into(dest, x, 0.85, 0.55, 1)
    ef, en, et = get_quadtree_features(x, 0.85, 0.55, 1)
    assert f.data_ptr() == dest.data_ptr() and torch.equal(f, ef) and torch.equal(n, en) and torch.equal(t, et)
    assert torch.all(dest[f.shape[0]:] == 7.0)                     # nothing beyond row N' is written
    # merged_token_1d_idx written by the kernels == the hook's arithmetic on tlbr (quadtree_attn_monkey_patch.py:103-104)
    f2, n2, t2, idx = get_quadtree_features_into(dest, x, 0.85, 0.55, 1, return_idx=True)
    assert idx.dtype == torch.int32 and torch.equal(idx, et[:, 0] * 196 + et[:, 1] * 14 + et[:, 2]) and torch.equal(t2, et)


def test_patched_qwen2_forward_runs_on_device_and_matches_oracle_glue():
    transformers = pytest.importorskip("transformers")
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    from oracle import sttm_oracle as O
    from sttm_amd import monkey_patch_interface as MPI
    from sttm_amd import patch_hooks
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    C, T = 64, 4
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval().to(dev)
    hs, start, length = _prompt(T, C, 14, 14, torch.float32)
    try:
        MPI.replace_qwen2_by_sparse_attn("quadtree", sa_start_layer_idx=1, sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.55,
                                         sa_tree_root_level=1)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(T)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
            # the same forward by hand, with the CPU oracle doing the merge on the layer-0 output
            pos = torch.arange(hs.shape[1], device=dev).unsqueeze(0)
            pe = model.rotary_emb(hs, pos)
            h = model.layers[0](hs, attention_mask=None, position_embeddings=pe, position_ids=pos)
            hm, pm, idx = patch_hooks.quadtree_merge_llava(h.cpu(), pos.cpu(), start, length, T, O.get_quadtree_features,
                                                           0.85, 0.55, 1, False)
            hm, pm = hm.to(dev), pm.to(dev)
            pe = model.rotary_emb(hm, pm)
            for layer in model.layers[1:]:
                hm = layer(hm, attention_mask=None, position_embeddings=pe, position_ids=pm)
            ref = model.norm(hm)
        assert torch.equal(model.merged_token_1d_idx.cpu().long(), idx.long())
        assert out.shape == ref.shape and out.shape[1] < hs.shape[1]
        assert torch.allclose(out, ref, atol=2e-5)
    finally:
        MPI.restore_qwen2()
        if "sttm_merge_fn" in Qwen2Model.__dict__:
            del Qwen2Model.sttm_merge_fn


@pytest.mark.parametrize("ver,weighted", [(1, False), (1, True), (2, False)])
def test_abl_pos_hook_on_device_matches_oracle_glue(ver, weighted):
    """quadtree-abl-pos hook (quadtree_attn_monkey_patch_for_abl_pos.py:88-136) through the HIP path vs the same glue with the oracle."""
    from oracle import sttm_oracle as O
    from sttm_amd import get_quadtree_features, patch_hooks
    T, C, H = 6, 64, 14
    hs, start, length = _prompt(T, C, H, H, torch.float32, seed=11)
    S = hs.shape[1]
    pos = torch.arange(S, device=hs.device).unsqueeze(0)
    g = torch.Generator().manual_seed(5)
    pe = tuple(torch.randn(1, S, 16, generator=g).to(hs.device) for _ in range(2))

    def rot(h, p):
        return tuple(torch.cos(p.float()).unsqueeze(-1).expand(-1, -1, 16) * (k + 1) for k in range(2))
    a = patch_hooks.quadtree_merge_abl_pos(hs, pos, pe, start, length, T, get_quadtree_features, 0.85, 0.55, 1, False, ver,
                                           weighted, rot)
    b = patch_hooks.quadtree_merge_abl_pos(hs.cpu(), pos.cpu(), tuple(p.cpu() for p in pe), start, length, T,
                                           O.get_quadtree_features, 0.85, 0.55, 1, False, ver, weighted, rot)
    assert torch.equal(a[3].cpu(), b[3]) and torch.equal(a[1].cpu(), b[1])
    assert float((a[0].cpu() - b[0]).abs().max()) <= 1e-5
    for u, v in zip(a[2], b[2]):
        assert float((u.cpu() - v).abs().max()) <= 1e-5


@pytest.mark.parametrize("pattern", ["quadtree", "dycoke-stage1", "tome"])
def test_patched_qwen2vl_text_model_runs_on_device(pattern):
    """Qwen2-VL text model (3-D mRoPE ids gathered by the merged-token index) through the HIP path, against the same forward done
    by hand with the CPU oracles doing the merge."""
    pytest.importorskip("transformers")
    try:
        from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLTextConfig
        from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextModel
    except Exception:  # noqa: BLE001
        pytest.skip("this transformers has no Qwen2VLTextModel")
    from oracle import dycoke_oracle as D
    from oracle import sttm_oracle as O
    from sttm_amd import monkey_patch_interface as MPI
    from sttm_amd import patch_hooks
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    C, T, H, W = 64, 6, 10, 18
    cfg = Qwen2VLTextConfig(vocab_size=64, hidden_size=C, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                            num_key_value_heads=2, max_position_embeddings=4096,
                            rope_parameters={"rope_type": "default", "mrope_section": [2, 2, 4], "rope_theta": 10000.0})
    cfg._attn_implementation = "sdpa"
    model = Qwen2VLTextModel(cfg).eval().to(dev)
    hs, start, length = _prompt(T, C, H, W, torch.float32, seed=21)
    S = hs.shape[1]
    pos = torch.stack([torch.arange(S), torch.arange(S) // 2, torch.arange(S) // 3]).unsqueeze(1).to(dev)     # [3, 1, S]
    kw = {"quadtree": dict(sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.6, sa_tree_root_level=1),
          "dycoke-stage1": dict(sa_prune_ratio=0.7), "tome": dict(sa_prune_ratio=0.5, sa_tome_ver="video")}[pattern]
    try:
        MPI.replace_qwen2_by_sparse_attn(pattern, sa_start_layer_idx=1, **kw)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(T)
        model.image_H = torch.tensor(H)
        model.image_W = torch.tensor(W)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, position_ids=pos, use_cache=False).last_hidden_state
            pe = model.rotary_emb(hs, pos)
            h = model.layers[0](hs, attention_mask=None, position_embeddings=pe, position_ids=None)
            hc, pc = h.cpu(), pos.cpu()
            if pattern == "quadtree":
                hm, p2, _, _ = patch_hooks.quadtree_merge_qwen2vl(hc, pc, start, length, T, H, W, O.get_quadtree_features, 0.85, 0.6, 1, False)
            elif pattern == "dycoke-stage1":
                hm, p2, _ = patch_hooks.dycoke_merge(hc, pc, start, length, T, D.dycoke_ttm, 0.7, gather_positions=True)
            else:
                # glue written out from token_merging_qwen2vl_monkey_patch/tome_attn_monkey_patch.py:96-108 (NOT patch_hooks):
                # system ++ merged ++ instruction, visual mRoPE ids gathered by the ToMe token index
                end = start + length
                video = hc[0, start:end].reshape(T, H, W, C).permute(0, 3, 1, 2)
                tf, tidx = O.get_tome_features(video, 0.5, "video")
                hm = torch.cat([hc[:, :start], tf.unsqueeze(0), hc[:, end:]], dim=1)
                p2 = torch.cat([pc[:, :, :start], pc[:, :, start:end][:, :, tidx], pc[:, :, end:]], dim=-1)
                # ratio 0.5 merges every even token into an odd one: the output order is exact
                assert torch.equal(model.merged_token_1d_idx.cpu(), tidx)
            hm, p2 = hm.to(dev), p2.to(dev)
            pe = model.rotary_emb(hm, p2)
            for layer in model.layers[1:]:
                hm = layer(hm, attention_mask=None, position_embeddings=pe, position_ids=None)
            ref = model.norm(hm)
        assert out.shape == ref.shape and out.shape[1] < S
        assert torch.allclose(out, ref, atol=2e-5)
    finally:
        MPI.restore_qwen2()


@pytest.mark.parametrize("pattern", ["tome", "dycoke-stage1", "quadtree"])
def test_hooks_on_a_bf16_model(pattern):
    """The production dtype: LLaVA-Video / Qwen2-VL checkpoints run bf16 hidden states (eval_vidqa_by_feat_llavavideo.py:104), and
    that is what the installed hooks hand to get_tome_features / dycoke_ttm / get_quadtree_features.  (a) the patched bf16
    Qwen2Model runs end to end on the GPU; (b) the hook's merge on the bf16 hidden states agrees with the oracle on the same
    tensor: ToMe ratio 0.5 keeps exactly the odd tokens (ids equal), DyCoke / quadtree ids as reported."""
    pytest.importorskip("transformers")
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    from oracle import dycoke_oracle as D
    from oracle import sttm_oracle as O
    from sttm_amd import monkey_patch_interface as MPI
    from sttm_amd import get_quadtree_features, get_tome_features, patch_hooks
    from sttm_amd.dycoke_merger import dycoke_ttm
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    C, T, H = 128, 6, 14
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval().to(dev, torch.bfloat16)
    hs, start, length = _prompt(T, C, H, H, torch.bfloat16, seed=31)
    pos = torch.arange(hs.shape[1], device=dev).unsqueeze(0)
    kw = {"quadtree": dict(sa_tree_thresh=0.85, sa_tree_temporal_thresh=0.55, sa_tree_root_level=1),
          "dycoke-stage1": dict(sa_prune_ratio=0.7), "tome": dict(sa_prune_ratio=0.5, sa_tome_ver="video")}[pattern]
    try:
        MPI.replace_qwen2_by_sparse_attn(pattern, sa_start_layer_idx=1, **kw)
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(length)
        model.num_frame = torch.tensor(T)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
        assert out.dtype == torch.bfloat16 and out.shape[1] < hs.shape[1] and bool(torch.isfinite(out.float()).all())
        # the merge itself, on the same bf16 tensor, device vs oracle
        if pattern == "tome":
            hm, _, tok = patch_hooks.tome_merge(hs, pos, start, length, T, get_tome_features, 0.5, "video")
            rm, _, rtok = patch_hooks.tome_merge(hs.cpu(), pos.cpu(), start, length, T, O.get_tome_features, 0.5, "video")
            assert torch.equal(tok.cpu(), rtok) and hm.shape == rm.shape
            rel = (hm.cpu().float() - rm.float()).abs() / rm.float().abs().clamp_min(1.0)
            assert float((rel > 2.0 ** -6).any(dim=-1).float().mean()) <= 0.03
            assert out.shape[1] == hm.shape[1]
        elif pattern == "dycoke-stage1":
            hm, _, idx = patch_hooks.dycoke_merge(hs, pos, start, length, T, dycoke_ttm, 0.7)
            rm, _, ridx = patch_hooks.dycoke_merge(hs.cpu(), pos.cpu(), start, length, T, D.dycoke_ttm, 0.7)
            assert hm.shape == rm.shape and out.shape[1] == hm.shape[1]
            got, ref = set(idx.cpu().tolist()), set(ridx.tolist())
            agree = len(got & ref) / len(ridx)
            print(f"dycoke-stage1 on bf16: kept-token agreement {agree:.4f}; only here {sorted(got - ref)[:8]}, only in the oracle {sorted(ref - got)[:8]}")
            assert agree >= 0.995                                       # measured 0.9984: bf16 similarities tie at the cut; see test_hip_dycoke.py
        else:
            hm, _, idx = patch_hooks.quadtree_merge_llava(hs, pos, start, length, T, get_quadtree_features, 0.85, 0.55, 1, False)
            rm, _, ridx = patch_hooks.quadtree_merge_llava(hs.cpu(), pos.cpu(), start, length, T, O.get_quadtree_features, 0.85, 0.55, 1, False)
            assert torch.equal(idx.cpu(), ridx) and out.shape[1] == hm.shape[1]
    finally:
        MPI.restore_qwen2()
