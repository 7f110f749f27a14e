"""GPU parity tests of the upstream 2-D pooling kernel (sttm_amd.upstream.get_2dPool -> sttm_pool2d) against the vectors the
reference's get_2dPool produced and against the CPU oracle at the production shape."""
import os

import pytest
import torch

from tests._golden import POOL_GOLDEN as GOLDEN, load_pool_case, pool_close_enough as close_enough

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_pool_golden_vectors(path):
    from sttm_amd.upstream import get_2dPool
    meta, x, y = load_pool_case(path)
    out = get_2dPool(x.to(DEV), stride=meta["stride"], width=meta.get("width", -1), mode=meta["mode"],
                     num_patches_per_side=meta["side"])
    assert out.shape[1] == meta["out_tokens"]
    assert close_enough(out.cpu(), y, meta)


@pytest.mark.parametrize("T,side,C,stride,mode,dtype", [
    (16, 27, 3584, 2, "bilinear", torch.bfloat16),       # LLaVA-Video: 27x27 SigLIP patches -> 14x14, Qwen2-7B width
    (8, 27, 1024, 2, "bilinear", torch.float32),
    (4, 27, 896, 2, "average", torch.float16),
    (4, 24, 1024, 3, "max", torch.bfloat16),
    (3, 27, 50, 2, "bilinear", torch.float32),           # narrow rows: 2-wide packs
    (3, 9, 7, 2, "average", torch.float32),              # odd channel count: scalar packs
])
def test_pool_matches_oracle_bit_exactly(T, side, C, stride, mode, dtype):
    """The kernel follows the oracle's arithmetic order (float32, one rounding per operation): results are identical."""
    from oracle import pool_oracle as P
    from sttm_amd.upstream import get_2dPool
    g = torch.Generator().manual_seed(T * 1000 + C)
    x = torch.randn(T, side * side, C, generator=g).to(dtype)
    exp = P.get_2dpool(x, stride, side, side, mode)
    out = get_2dPool(x.to(DEV), stride=stride, width=side, mode=mode)
    assert out.dtype == dtype and out.shape == exp.shape
    assert torch.equal(out.cpu(), exp)


def test_pool_interface_behaviour():
    from sttm_amd.upstream import get_2dPool
    x = torch.randn(2, 81, 8, device=DEV)
    assert get_2dPool(x, stride=1) is x                                  # llava_arch.py:174-175
    with pytest.raises(ValueError, match="Unexpected mm_spatial_pool_mode"):
        get_2dPool(x, stride=2, mode="nearest")
    with pytest.raises(RuntimeError):
        get_2dPool(x, stride=2, width=10)                                # 10*10 != 81: the reference's view() fails too
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_2dPool(x.cpu(), stride=2)


def test_pool_then_merge_pipeline_shapes():
    """27x27 projector tokens -> 14x14 -> STTM merge: the producer and the hot path chained on the device."""
    from sttm_amd import get_quadtree_features
    from sttm_amd.upstream import get_2dPool
    T, C = 8, 256
    g = torch.Generator().manual_seed(5)
    base = torch.randn(1, 729, C, generator=g)
    x = (base + 0.05 * torch.randn(T, 729, C, generator=g)).to(DEV)      # near-static clip: the temporal stage merges a lot
    pooled = get_2dPool(x, stride=2, mode="bilinear", num_patches_per_side=27)
    assert pooled.shape == (T, 196, C)
    video = pooled.reshape(T, 14, 14, C).permute(0, 3, 1, 2)
    f, n, t = get_quadtree_features(video, 0.85, 0.55, 1)
    assert f.shape[0] == n.shape[0] == t.shape[0] and 0 < f.shape[0] < T * 196
    assert int(n.sum()) == T * 196


@pytest.mark.parametrize("T,side,C,tgt,dtype", [(8, 14, 1024, 10, torch.float32), (6, 14, 3584, 7, torch.bfloat16),
                                               (4, 27, 256, 14, torch.float16), (3, 10, 34, 13, torch.float32)])
def test_resize_nearest_matches_oracle_exactly(T, side, C, tgt, dtype):
    from oracle import pool_oracle as P
    from sttm_amd.upstream import resize_nearest
    g = torch.Generator().manual_seed(tgt)
    x = torch.randn(T, side * side, C, generator=g).to(dtype)
    out = resize_nearest(x.to(DEV), side, side, (tgt, tgt))
    assert torch.equal(out.cpu(), P.resize_nearest(x, side, side, (tgt, tgt)))


def test_pyrd_pattern_runs_on_device():
    transformers = pytest.importorskip("transformers")
    import torch.nn.functional as F
    from transformers import Qwen2Config
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Model
    from sttm_amd import monkey_patch_interface as MPI
    torch.manual_seed(0)
    C, T, start = 64, 4, 5
    cfg = Qwen2Config(vocab_size=64, hidden_size=C, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=4096, attn_implementation="sdpa")
    model = Qwen2Model(cfg).eval().to(DEV)
    hs = torch.randn(1, start + T * 196 + 9, C, device=DEV)
    try:
        MPI.replace_qwen2_by_sparse_attn("pyrd", sa_pyrd_loc_list=[1], sa_pyrd_size_list=[10])
        model.image_token_start_index = torch.tensor(start)
        model.image_token_length = torch.tensor(T * 196)
        model.num_frame = torch.tensor(T)
        with torch.inference_mode():
            out = model(inputs_embeds=hs, use_cache=False).last_hidden_state
            pos = torch.arange(hs.shape[1], device=DEV).unsqueeze(0)
            pe = model.rotary_emb(hs, pos)
            h = model.layers[0](hs, attention_mask=None, position_embeddings=pe, position_ids=pos)
            vis = h[0, start:start + T * 196].reshape(T, 14, 14, C).permute(0, 3, 1, 2)
            r = F.interpolate(vis, size=(10, 10)).permute(0, 2, 3, 1).reshape(1, T * 100, C)
            h = torch.cat([h[:, :start], r, h[:, start + T * 196:]], dim=1)
            pos = pos[:, :h.shape[1]]
            pe = model.rotary_emb(h, pos)
            for layer in model.layers[1:]:
                h = layer(h, attention_mask=None, position_embeddings=pe, position_ids=pos)
            ref = model.norm(h)
        assert out.shape == ref.shape and torch.allclose(out, ref, atol=2e-5)
    finally:
        MPI.restore_qwen2()


def test_feature_files_of_the_reference_feed_the_merge_path(tmp_path):
    """The on-disk formats of the reference's extraction scripts (video_feat_llavavideo.py:89-95: [T, 729, 1152] bf16 before the
    projector; video_feat_qwen2vl.py:72-79: [T, H, W, C]) -> loader -> projector (caller's) -> HIP 27->14 pooling -> the merge
    path, against the same steps in plain torch + the oracle."""
    import torch.nn.functional as F
    from oracle import sttm_oracle as O
    from sttm_amd import get_quadtree_features
    from sttm_amd.upstream import load_llavavideo_features, load_qwen2vl_features
    g = torch.Generator().manual_seed(3)
    T, Cv, C = 6, 96, 128
    tower = (torch.randn(T, 1, Cv, generator=g) * 0.5 + torch.randn(1, 729, Cv, generator=g) + 0.3 * torch.randn(T, 729, Cv, generator=g)).bfloat16()
    path = tmp_path / "vid0.pt"
    torch.save(tower, path)
    proj = torch.nn.Linear(Cv, C).to(DEV, torch.bfloat16)
    with torch.no_grad():
        video, side = load_llavavideo_features(str(path), DEV, projector=proj)
        assert side == 14 and video.shape == (T, C, 14, 14) and video.dtype == torch.bfloat16 and video.stride(1) == 1
        # the reference's steps in torch: project, NCHW bilinear resize to ceil(27 / 2) = 14, back to tokens
        ref = proj(tower.to(DEV)).view(T, 27, 27, C).permute(0, 3, 1, 2)
        ref = F.interpolate(ref.float(), size=(14, 14), mode="bilinear").to(torch.bfloat16)
        d = (video.float() - ref.float()).abs() / ref.float().abs().clamp_min(1.0)
        assert float(d.max()) <= 2 ** -7                                    # one bf16 ulp (the fp32 taps are rounded once either way)
        out = get_quadtree_features(video, 0.85, 0.55, 1)
        exp = O.get_quadtree_features(video.cpu(), 0.85, 0.55, 1)
        assert torch.equal(out[2].cpu(), exp[2]) and torch.equal(out[1].cpu(), exp[1])
    grid = torch.randn(5, 10, 18, 64, generator=g).bfloat16()
    path2 = tmp_path / "vid1.pt"
    torch.save(grid, path2)
    v2, (t2, h2, w2) = load_qwen2vl_features(str(path2), DEV)
    assert (t2, h2, w2) == (5, 10, 18) and v2.shape == (5, 64, 10, 18) and v2.stride(1) == 1
    assert torch.equal(v2.permute(0, 2, 3, 1).cpu().view(torch.int16), grid.view(torch.int16))
    out = get_quadtree_features(v2, 0.85, 0.6, 1)
    exp = O.get_quadtree_features(v2.cpu(), 0.85, 0.6, 1)
    assert torch.equal(out[2].cpu(), exp[2])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        load_qwen2vl_features(str(path2), "cpu")


# ---- get_2dPool fused into the spatial kernel's leaf load (sttm_quadtree_merge_pooled) -------------------------------

FUSED_POOL_CASES = [
    # (T, side, C, stride, mode, dtype, spatial thr, temporal thr, root_level, weighted, slow)
    (8, 27, 1024, 2, "bilinear", torch.float32, 0.85, 0.55, 1, False, False),     # 27 x 27 -> 14 x 14, the LLaVA-Video layout
    (6, 27, 3584, 2, "bilinear", torch.bfloat16, 0.85, 0.55, 1, False, False),    # production width (7B), bf16 hidden states
    (5, 27, 512, 2, "bilinear", torch.float16, 0.80, 0.50, 1, False, False),
    (6, 27, 256, 2, "bilinear", torch.float32, 0.85, -1.0, 1, False, False),      # spatial stage only
    (6, 27, 256, 2, "bilinear", torch.float32, 0.80, 0.50, 1, True, False),       # weighted_avg (sum-pool pyramid over pooled leaves)
    (6, 27, 256, 2, "bilinear", torch.float32, 0.85, 0.55, 1, False, True),       # slow_ver
    (5, 28, 256, 2, "average", torch.float32, 0.85, 0.55, 1, False, False),       # 2 x 2 average window
    (5, 28, 512, 2, "max", torch.bfloat16, 0.85, 0.55, 1, False, False),          # 2 x 2 max window
    (4, 40, 256, 3, "bilinear", torch.float32, 0.85, 0.55, 1, False, False),      # stride 3: 40 -> 14
    (4, 25, 128, 2, "bilinear", torch.bfloat16, 0.85, 0.60, 1, False, False),     # 25 -> 13 x 13 (odd pooled grid)
]


def _smooth_tokens(T, side, C, seed, dtype):
    """projector-like tokens with spatial and temporal redundancy (so that nodes of every size and cross-frame merges occur)"""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, 1, 1, C, generator=g)
    coarse = torch.randn(T, (side + 7) // 8, (side + 7) // 8, C, generator=g) * 0.6
    coarse = coarse.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :side, :side]
    drift = torch.randn(T, 1, 1, C, generator=g).cumsum(0) * 0.05
    x = base + coarse + drift + 0.25 * torch.randn(T, side, side, C, generator=g)
    return x.reshape(T, side * side, C).to(dtype)


@pytest.mark.parametrize("case", FUSED_POOL_CASES, ids=lambda c: "T%d_s%d_C%d_st%d_%s" % c[:5])
def test_fused_pooled_input_equals_pool_then_merge(case):
    """`get_quadtree_features_from_pooled_input` (the pool fused into K1's leaf load) against (a) pool_oracle o sttm_oracle on the CPU and
    (b) the device's own two-step form: identical indices, features bit-identical to the two-step form."""
    from oracle import pool_oracle as P
    from oracle import sttm_oracle as O
    from sttm_amd import get_quadtree_features, get_quadtree_features_from_pooled_input
    from sttm_amd.upstream import get_2dPool
    T, side, C, stride, mode, dtype, thr, tthr, root, weighted, slow = case
    x = _smooth_tokens(T, side, C, 300 + T + C, dtype)
    pooled = P.get_2dpool(x, stride, side, side, mode)
    hw = int(round(pooled.shape[1] ** 0.5))
    ef, en, et = O.get_quadtree_features(pooled.reshape(T, hw, hw, C).permute(0, 3, 1, 2), thr, tthr, root, weighted, slow_ver=slow)
    f, n, t = get_quadtree_features_from_pooled_input(x.to(DEV), thr, tthr, root, weighted, slow, stride=stride, mode=mode, width=side, force_fused=True)
    assert torch.equal(t.cpu(), et) and torch.equal(n.cpu(), en)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((f.float().cpu() - ef.float()).abs().max()) <= tol
    dp = get_2dPool(x.to(DEV), stride=stride, width=side, mode=mode)
    f2, n2, t2 = get_quadtree_features(dp.reshape(T, hw, hw, C).permute(0, 3, 1, 2), thr, tthr, root, weighted, slow_ver=slow)
    assert torch.equal(t, t2) and torch.equal(n, n2) and torch.equal(f, f2)
    assert 0 < f.shape[0] < T * hw * hw                       # the case really merges something


@pytest.mark.parametrize("path", [p for p in GOLDEN if "bilinear_27_s2" in os.path.basename(p)], ids=os.path.basename)
def test_fused_pooled_input_on_the_reference_generated_pool_vectors(path):
    """The `pool_bilinear_27_s2*` vectors (inputs + what the REFERENCE's get_2dPool returned): the fused call on the input equals the
    merge of the reference's pooled output."""
    from oracle import sttm_oracle as O
    from sttm_amd import get_quadtree_features_from_pooled_input
    meta, x, y = load_pool_case(path)
    T, _, C = x.shape
    hw = int(round(y.shape[1] ** 0.5))
    for thr, tthr in ((0.85, 0.55), (0.6, 0.4)):
        ef, en, et = O.get_quadtree_features(y.reshape(T, hw, hw, C).permute(0, 3, 1, 2), thr, tthr, 1)
        f, n, t = get_quadtree_features_from_pooled_input(x.to(DEV), thr, tthr, 1, stride=meta["stride"], mode=meta["mode"],
                                                          num_patches_per_side=meta["side"], force_fused=True)
        if (C * x.element_size()) % 16 == 0:                 # (narrow vectors take the two-step form: same check)
            pass
        assert torch.equal(t.cpu(), et) and torch.equal(n.cpu(), en)
        assert float((f.float().cpu() - ef.float()).abs().max()) <= (1e-5 if x.dtype == torch.float32 else 2e-2)


def test_fused_pooled_input_falls_back_and_rejects_like_get_2dpool():
    from oracle import pool_oracle as P
    from oracle import sttm_oracle as O
    from sttm_amd import _lib, get_quadtree_features_from_pooled_input
    x = _smooth_tokens(4, 27, 96, 9, torch.float32)
    # root_level 0 on 14 x 14 is a 4-level tree, head_dim is the per-head cosine: both take pool2d + merge on the device
    for kw in (dict(root_level=0), dict(root_level=1, head_dim=32)):
        pooled = P.get_2dpool(x, 2, 27, 27, "bilinear").reshape(4, 14, 14, 96).permute(0, 3, 1, 2)
        ef, en, et = O.get_quadtree_features(pooled, 0.85, 0.55, kw["root_level"], head_dim=kw.get("head_dim"))
        f, n, t = get_quadtree_features_from_pooled_input(x.to(DEV), 0.85, 0.55, kw["root_level"], head_dim=kw.get("head_dim"))
        assert torch.equal(t.cpu(), et) and torch.equal(n.cpu(), en)
    with pytest.raises(ValueError, match="Unexpected mm_spatial_pool_mode"):
        get_quadtree_features_from_pooled_input(x.to(DEV), 0.85, mode="nearest")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_quadtree_features_from_pooled_input(x, 0.85)
    # the C ABI says UNSUPPORTED for what the fused kernel does not cover (and nothing is launched)
    lib = _lib.load()
    d = x.to(DEV)
    ws = torch.empty(lib.sttm_quadtree_workspace_bytes(4, 14, 14, 96, 0, 0), dtype=torch.uint8, device=DEV)
    out = torch.empty(4 * 196, 96, device=DEV); npc = torch.empty(4 * 196, dtype=torch.int32, device=DEV)
    tl = torch.empty(4 * 196, 5, dtype=torch.int32, device=DEV); cnt = torch.zeros(8, dtype=torch.int32, device=DEV)
    args = lambda root, mode, stride: (d.data_ptr(), 4, 27, 27, 96, 0, mode, stride, 0.85, 0.55, root, 0, 0, ws.data_ptr(), ws.numel(),   # noqa: E731
                                       out.data_ptr(), npc.data_ptr(), tl.data_ptr(), cnt.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream, 0)
    assert lib.sttm_quadtree_merge_pooled(*args(0, 2, 2)) == _lib.ERR_UNSUPPORTED          # 4-level tree
    assert lib.sttm_quadtree_merge_pooled(*args(1, 0, 3)) == _lib.ERR_UNSUPPORTED          # 3 x 3 average window
    assert lib.sttm_quadtree_merge_pooled(*args(1, 2, 1)) == _lib.ERR_ARG                  # stride 1 = identity
    assert lib.sttm_quadtree_merge_pooled(*args(1, 7, 2)) == _lib.ERR_ARG                  # unknown mode
    assert lib.sttm_quadtree_merge_pooled(*args(1, 2, 2)) == 0
    torch.cuda.synchronize()
