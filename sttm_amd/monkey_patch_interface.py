"""L2 installer with the reference's names and keyword arguments (token_merging_utils/monkey_patch_interface.py:17-38).

`replace_qwen2_by_sparse_attn(pattern_name, **kwargs)` dispatches on the same pattern names.  "quadtree" and "tome" --
the two patterns on the hot path -- and the baselines that share their hook ("quadtree-abl-pos", "pyrd", "dycoke-stage1",
"octree") are implemented: like the reference they store their configuration as CLASS attributes on the decoder model
class and overwrite its `.forward` (quadtree_attn_monkey_patch.py:177-187, tome_attn_monkey_patch.py:163-171).  Only full
DyCoke (decode-time KV pruning), the FrameFusion baselines and the plotting variants raise NotImplementedError, with a message
that says so.

The reference pins transformers==4.45.2 and patches a verbatim copy of that version's Qwen2Model.forward.  This
build targets the transformers that is installed (5.x decoder-layer API): the patched forward below is the
installed `Qwen2Model.forward` with the merge step inserted before layer `sa_start_layer_idx` on prefill.
State read by the hook is the same: `self.image_token_start_index`, `self.image_token_length`, `self.num_frame`
(0-d tensors set by generate(), llava/model/language_model/llava_qwen.py:141-143), plus `image_H/image_W` for Qwen2-VL.
"""
import torch

from . import patch_hooks
from .quadtree_interface import get_quadtree_features, get_quadtree_features_into
from .tome_interface import get_tome_features

_UNIMPLEMENTED = {
    "quadtree_vis": "visualisation variant",
    "dycoke": "DyCoke with stage-2 KV-cache pruning during decoding",
}


def _item(v):
    return int(v.item()) if torch.is_tensor(v) else int(v)


def _is_prefill(past_key_values):
    return past_key_values is None or past_key_values.get_seq_length() == 0


def _rebuild_masks(self, mask_map, hidden_states, past_key_values, position_ids=None):
    """Masks for the SHORTER sequence after the merge.  The reference computes `causal_mask` once before the layer loop
    and never updates it (quadtree_attn_monkey_patch.py:60-66 vs :88-117), which only works when that mask is None
    (flash-attention / SDPA without padding).  Here the masks are rebuilt on the merged hidden states, so eager attention
    stays causal and sliding-window layers keep their window; for SDPA / flash this returns None exactly like before."""
    from transformers.masking_utils import create_causal_mask, create_sliding_window_causal_mask
    # prefill: the layers from here on see an empty cache, so keys = the merged sequence itself (no past offset: the
    # cache object already holds the LONG sequence of the earlier layers and must not be consulted); no position ids
    # either -- gathered (non-monotonic) ids would be mistaken for packed sequences
    mk = dict(config=self.config, inputs_embeds=hidden_states, attention_mask=None, past_key_values=None, position_ids=None)
    out = {}
    for k in mask_map:
        out[k] = create_sliding_window_causal_mask(**mk) if k == "sliding_attention" else create_causal_mask(**mk)
    return out


def _qwen2_forward_with_merge(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                              inputs_embeds=None, use_cache=None, **kwargs):
    """transformers 5.x Qwen2Model.forward + the STTM / ToMe hook (prefill, batch 1)."""
    from transformers.cache_utils import DynamicCache
    from transformers.masking_utils import create_causal_mask, create_sliding_window_causal_mask
    from transformers.modeling_outputs import BaseModelOutputWithPast
    if (input_ids is None) ^ (inputs_embeds is not None):
        raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
    if inputs_embeds is None:
        inputs_embeds = self.embed_tokens(input_ids)
    if use_cache and past_key_values is None:
        past_key_values = DynamicCache(config=self.config)
    prefilling = _is_prefill(past_key_values)
    if position_ids is None:
        seen = past_key_values.get_seq_length() if past_key_values is not None else 0
        position_ids = (torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + seen).unsqueeze(0)
    if not isinstance(mask_map := attention_mask, dict):
        mk = dict(config=self.config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                  past_key_values=past_key_values, position_ids=position_ids)
        mask_map = {"full_attention": create_causal_mask(**mk)}
        if getattr(self, "has_sliding_layers", False):
            mask_map["sliding_attention"] = create_sliding_window_causal_mask(**mk)
    hidden_states = inputs_embeds
    position_embeddings = self.rotary_emb(hidden_states, position_ids)
    merged = False
    for i, layer in enumerate(self.layers[: self.config.num_hidden_layers]):
        if (prefilling and self.sttm_pattern == "pyrd" and i in self.sa_pyrd_llm_idxs
                and getattr(self, "image_token_length", None) is not None):
            # fixed-size pyramid baseline: may fire at several layers, each shrinking the frames further
            start, length, T = _item(self.image_token_start_index), _item(self.image_token_length), _item(self.num_frame)
            hidden_states, position_ids, new_len = patch_hooks.pyrd_resize(
                hidden_states, position_ids, start, length, T, self.sa_pyrd_idx2size[i], type(self).sttm_resize_fn)
            self.image_token_length = torch.tensor(new_len)              # like the reference (:101): persists on the module
            position_embeddings = self.rotary_emb(hidden_states, position_ids)
            mask_map = _rebuild_masks(self, mask_map, hidden_states, past_key_values, position_ids)
        elif (prefilling and not merged and self.sttm_pattern != "pyrd" and i == self.sa_start_layer_idx
                and getattr(self, "image_token_length", None) is not None):
            start, length, T = _item(self.image_token_start_index), _item(self.image_token_length), _item(self.num_frame)
            if self.sttm_pattern == "quadtree-abl-pos":
                head_dim = layer.self_attn.head_dim if self.sim_per_head else None
                hidden_states, position_ids, position_embeddings, idx = patch_hooks.quadtree_merge_abl_pos(
                    hidden_states, position_ids, position_embeddings, start, length, T, type(self).sttm_merge_fn,
                    self.sa_tree_thresh, self.sa_tree_temporal_thresh, self.sa_tree_root_level, self.sa_tree_weighted_avg,
                    self.pos_emb_ver, self.pos_emb_weighted_avg, self.rotary_emb, slow_ver=self.sttm_slow_ver, head_dim=head_dim)
            elif self.sttm_pattern == "quadtree":
                head_dim = layer.self_attn.head_dim if self.sim_per_head else None
                hidden_states, position_ids, idx = patch_hooks.quadtree_merge_llava(
                    hidden_states, position_ids, start, length, T, type(self).sttm_merge_fn,
                    self.sa_tree_thresh, self.sa_tree_temporal_thresh, self.sa_tree_root_level, self.sa_tree_weighted_avg,
                    slow_ver=self.sttm_slow_ver, head_dim=head_dim,
                    merge_into_fn=_fused_merge_fn(type(self)))
            elif self.sttm_pattern == "dycoke-stage1":
                hidden_states, position_ids, idx = patch_hooks.dycoke_merge(
                    hidden_states, position_ids, start, length, T, type(self).sttm_dycoke_fn, self.sa_prune_ratio)
            elif self.sttm_pattern == "octree":
                hidden_states, position_ids = patch_hooks.octree_merge(
                    hidden_states, position_ids, start, length, T, type(self).sttm_octree_fn, self.sa_tree_thresh,
                    self.sa_tree_root_level)
                idx = None                                            # the reference's octree hook keeps no token index
            else:
                hidden_states, position_ids, idx = patch_hooks.tome_merge(
                    hidden_states, position_ids, start, length, T, type(self).sttm_tome_fn, self.sa_prune_ratio, self.sa_tome_ver)
            self.merged_token_1d_idx = idx
            if self.sttm_pattern != "quadtree-abl-pos":               # that hook decides its own position embeddings
                position_embeddings = self.rotary_emb(hidden_states, position_ids)
            # batch-1 prefill without padding: the shorter sequence is plain causal
            mask_map = _rebuild_masks(self, mask_map, hidden_states, past_key_values, position_ids)
            merged = True
        hidden_states = layer(hidden_states, attention_mask=mask_map[self.config.layer_types[i]],
                              position_embeddings=position_embeddings, position_ids=position_ids,
                              past_key_values=past_key_values, use_cache=use_cache, **kwargs)
    hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=past_key_values if use_cache else None)


def _qwen2vl_forward_with_merge(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                                inputs_embeds=None, use_cache=None, **kwargs):
    """transformers 5.x Qwen2VLTextModel.forward + the STTM / ToMe hook: 3-D mRoPE position ids are GATHERED by the
    merged-token index (token_merging_qwen2vl_monkey_patch/quadtree_attn_monkey_patch.py:109-113); H and W come from
    `self.image_H / self.image_W` (llava/model/qwen2vl/modeling_qwen2vl.py:1919-1920)."""
    from transformers.cache_utils import DynamicCache
    from transformers.masking_utils import create_causal_mask, create_sliding_window_causal_mask
    from transformers.modeling_outputs import BaseModelOutputWithPast
    if (input_ids is None) ^ (inputs_embeds is not None):
        raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
    if use_cache and past_key_values is None:
        past_key_values = DynamicCache(config=self.config)
    if inputs_embeds is None:
        inputs_embeds = self.embed_tokens(input_ids)
    prefilling = _is_prefill(past_key_values)
    if position_ids is None:
        seen = past_key_values.get_seq_length() if past_key_values is not None else 0
        position_ids = torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + seen
        position_ids = position_ids.view(1, 1, -1).expand(3, inputs_embeds.shape[0], -1)
    elif position_ids.ndim == 2:
        position_ids = position_ids[None, ...].expand(3, position_ids.shape[0], -1)
    text_position_ids = None
    if position_ids.ndim == 3 and position_ids.shape[0] == 4:
        text_position_ids = position_ids[0]
        position_ids = position_ids[1:]
    if not isinstance(mask_map := attention_mask, dict):
        mk = dict(config=self.config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                  past_key_values=past_key_values, position_ids=text_position_ids)
        mask_map = {"full_attention": create_causal_mask(**mk)}
        if getattr(self, "has_sliding_layers", False):
            mask_map["sliding_attention"] = create_sliding_window_causal_mask(**mk)
    hidden_states = inputs_embeds
    position_embeddings = self.rotary_emb(hidden_states, position_ids)
    merged = False
    for i, layer in enumerate(self.layers):
        if prefilling and not merged and i == self.sa_start_layer_idx and getattr(self, "image_token_length", None) is not None:
            start, length, T = _item(self.image_token_start_index), _item(self.image_token_length), _item(self.num_frame)
            H, W = _item(self.image_H), _item(self.image_W)
            if self.sttm_pattern == "quadtree":
                hidden_states, position_ids, _cache_pos, idx = patch_hooks.quadtree_merge_qwen2vl(
                    hidden_states, position_ids, start, length, T, H, W, type(self).sttm_merge_fn,
                    self.sa_tree_thresh, self.sa_tree_temporal_thresh, self.sa_tree_root_level, self.sa_tree_weighted_avg,
                    slow_ver=self.sttm_slow_ver, merge_into_fn=_fused_merge_fn(type(self)))
            elif self.sttm_pattern == "dycoke-stage1":
                hidden_states, position_ids, idx = patch_hooks.dycoke_merge(
                    hidden_states, position_ids, start, length, T, type(self).sttm_dycoke_fn, self.sa_prune_ratio,
                    gather_positions=True)
            else:
                hidden_states, position_ids, idx = patch_hooks.tome_merge(
                    hidden_states, position_ids, start, length, T, type(self).sttm_tome_fn, self.sa_prune_ratio,
                    self.sa_tome_ver, H=H, W=W, gather_positions=True)
            if text_position_ids is not None:
                text_position_ids = text_position_ids[..., :hidden_states.size(1)]
            self.merged_token_1d_idx = idx
            position_embeddings = self.rotary_emb(hidden_states, position_ids)
            mask_map = _rebuild_masks(self, mask_map, hidden_states, past_key_values, text_position_ids)
            merged = True
        hidden_states = layer(hidden_states, attention_mask=mask_map[self.config.layer_types[i]],
                              position_embeddings=position_embeddings, position_ids=text_position_ids,
                              past_key_values=past_key_values, use_cache=use_cache, **kwargs)
    hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=past_key_values)


def _qwen2_model_class():
    import transformers.models.qwen2.modeling_qwen2 as m
    return m.Qwen2Model


def _qwen2vl_model_class():
    try:
        import transformers.models.qwen2_vl.modeling_qwen2_vl as m
        return getattr(m, "Qwen2VLTextModel", None) or getattr(m, "Qwen2VLModel", None)
    except Exception:  # noqa: BLE001
        return None


def _fused_merge_fn(cls):
    """Fused slice -> merge -> concat (the kernels write the merged rows into the new hidden-state buffer) is used
    whenever the merge function is the library's own; an injected `sttm_merge_fn` (tests) takes the plain path."""
    return get_quadtree_features_into if cls.sttm_merge_fn is get_quadtree_features else None


def replace_qwen2_with_quadtree_attn(sa_start_layer_idx=0, sa_tree_thresh=0.90, sa_tree_temporal_thresh=-1.0,
                                     sa_tree_root_level=0, sa_tree_weighted_avg=False, sttm_slow_ver=False,
                                     sim_per_head=False, **kwargs):
    print("Replace Qwen2 attention path by QuadTree (STTM) token merging [sttm_amd / MI355X]")
    cls = _qwen2_model_class()
    cls.sttm_pattern = "quadtree"
    cls.sa_start_layer_idx = sa_start_layer_idx
    cls.sa_tree_thresh = sa_tree_thresh
    cls.sa_tree_temporal_thresh = sa_tree_temporal_thresh
    cls.sa_tree_root_level = sa_tree_root_level
    cls.sa_tree_weighted_avg = sa_tree_weighted_avg
    cls.sttm_slow_ver = sttm_slow_ver
    cls.sim_per_head = sim_per_head
    if not hasattr(cls, "sttm_merge_fn"):
        cls.sttm_merge_fn = staticmethod(get_quadtree_features)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = _qwen2_forward_with_merge


def replace_qwen2_with_quadtree_attn_for_abl_pos(sa_start_layer_idx=0, sa_tree_thresh=0.90, sa_tree_temporal_thresh=-1.0,
                                                 sa_tree_root_level=0, sa_tree_weighted_avg=False, sttm_slow_ver=False,
                                                 sim_per_head=False, pos_emb_ver=0, pos_emb_weighted_avg=False, **kwargs):
    """quadtree_attn_monkey_patch_for_abl_pos.py:193-205."""
    replace_qwen2_with_quadtree_attn(sa_start_layer_idx, sa_tree_thresh, sa_tree_temporal_thresh, sa_tree_root_level,
                                     sa_tree_weighted_avg, sttm_slow_ver, sim_per_head)
    cls = _qwen2_model_class()
    cls.sttm_pattern = "quadtree-abl-pos"
    cls.pos_emb_ver = pos_emb_ver
    cls.pos_emb_weighted_avg = pos_emb_weighted_avg


def replace_qwen2_with_octree_attn(sa_start_layer_idx=0, sa_tree_thresh=0.90, sa_tree_root_level=0, **kwargs):
    """octree_attn_monkey_patch.py:163-168."""
    print("Replace Qwen2 attention path by OcTree token merging [sttm_amd / MI355X]")
    cls = _qwen2_model_class()
    cls.sttm_pattern = "octree"
    cls.sa_start_layer_idx = sa_start_layer_idx
    cls.sa_tree_thresh = sa_tree_thresh
    cls.sa_tree_root_level = sa_tree_root_level
    if not hasattr(cls, "sttm_octree_fn"):
        from .octree_utils import get_octree_features
        cls.sttm_octree_fn = staticmethod(get_octree_features)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = _qwen2_forward_with_merge


def _install_dycoke(cls, forward, sa_start_layer_idx, sa_prune_ratio):
    cls.sttm_pattern = "dycoke-stage1"
    cls.sa_start_layer_idx = sa_start_layer_idx
    cls.sa_prune_ratio = sa_prune_ratio
    if not hasattr(cls, "sttm_dycoke_fn"):
        from .dycoke_merger import dycoke_ttm
        cls.sttm_dycoke_fn = staticmethod(dycoke_ttm)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = forward


def replace_qwen2_with_dycoke_stage1_attn(sa_start_layer_idx=0, sa_prune_ratio=0.7, **kwargs):
    """dycoke_stage1_attn_monkey_patch.py:165-169."""
    print("Replace Qwen2 attention path by DyCoke stage-1 pruning [sttm_amd / MI355X]")
    _install_dycoke(_qwen2_model_class(), _qwen2_forward_with_merge, sa_start_layer_idx, sa_prune_ratio)


def replace_qwen2vl_with_dycoke_stage1_attn(sa_start_layer_idx=0, sa_prune_ratio=0.7, **kwargs):
    """token_merging_qwen2vl_monkey_patch/dycoke_stage1_attn_monkey_patch.py:165-168."""
    cls = _qwen2vl_model_class()
    if cls is None:
        return
    _install_dycoke(cls, _qwen2vl_forward_with_merge, sa_start_layer_idx, sa_prune_ratio)


def replace_qwen2_with_pyrd_attn(sa_pyrd_loc_list=[2], sa_pyrd_size_list=[10], **kwargs):
    """pyrd_attn_monkey_patch.py:167-173: resize every frame to size x size (nearest) before the listed decoder layers."""
    print("Replace Qwen2 attention path by Pyramid token merging [sttm_amd / MI355X]")
    assert len(sa_pyrd_loc_list) == len(sa_pyrd_size_list)
    cls = _qwen2_model_class()
    cls.sttm_pattern = "pyrd"
    cls.sa_pyrd_llm_idxs = list(sa_pyrd_loc_list)
    cls.sa_pyrd_idx2size = {sa_pyrd_loc_list[i]: sa_pyrd_size_list[i] for i in range(len(sa_pyrd_loc_list))}
    if not hasattr(cls, "sttm_resize_fn"):
        from .upstream import resize_nearest
        cls.sttm_resize_fn = staticmethod(resize_nearest)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = _qwen2_forward_with_merge


def replace_qwen2_with_tome_attn(sa_start_layer_idx=0, sa_prune_ratio=0.50, sa_tome_ver="frame", **kwargs):
    print("Replace Qwen2 attention path by ToMe token merging [sttm_amd / MI355X]")
    cls = _qwen2_model_class()
    cls.sttm_pattern = "tome"
    cls.sa_start_layer_idx = sa_start_layer_idx
    cls.sa_prune_ratio = sa_prune_ratio
    cls.sa_tome_ver = sa_tome_ver
    if not hasattr(cls, "sttm_tome_fn"):
        cls.sttm_tome_fn = staticmethod(get_tome_features)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = _qwen2_forward_with_merge


def replace_qwen2vl_with_quadtree_attn(sa_start_layer_idx=0, sa_tree_thresh=0.90, sa_tree_temporal_thresh=-1.0,
                                       sa_tree_root_level=0, sa_tree_weighted_avg=False, sttm_slow_ver=False, **kwargs):
    """Qwen2-VL text model (transformers 5.x `Qwen2VLTextModel`): same class-attribute mechanism, forward replaced by
    `_qwen2vl_forward_with_merge` (3-D position-id gather)."""
    cls = _qwen2vl_model_class()
    if cls is None:
        return
    print("Replace Qwen2-VL attention path by QuadTree (STTM) token merging [sttm_amd / MI355X]")
    cls.sttm_pattern = "quadtree"
    cls.sa_start_layer_idx = sa_start_layer_idx
    cls.sa_tree_thresh = sa_tree_thresh
    cls.sa_tree_temporal_thresh = sa_tree_temporal_thresh
    cls.sa_tree_root_level = sa_tree_root_level
    cls.sa_tree_weighted_avg = sa_tree_weighted_avg
    cls.sttm_slow_ver = sttm_slow_ver
    if not hasattr(cls, "sttm_merge_fn"):
        cls.sttm_merge_fn = staticmethod(get_quadtree_features)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = _qwen2vl_forward_with_merge


def replace_qwen2vl_with_tome_attn(sa_start_layer_idx=0, sa_prune_ratio=0.50, sa_tome_ver="frame", **kwargs):
    cls = _qwen2vl_model_class()
    if cls is None:
        return
    print("Replace Qwen2-VL attention path by ToMe token merging [sttm_amd / MI355X]")
    cls.sttm_pattern = "tome"
    cls.sa_start_layer_idx = sa_start_layer_idx
    cls.sa_prune_ratio = sa_prune_ratio
    cls.sa_tome_ver = sa_tome_ver
    if not hasattr(cls, "sttm_tome_fn"):
        cls.sttm_tome_fn = staticmethod(get_tome_features)
    if not hasattr(cls, "_sttm_original_forward"):
        cls._sttm_original_forward = cls.forward
    cls.forward = _qwen2vl_forward_with_merge


def restore_qwen2():
    """Undo replace_qwen2[vl]_with_* (not in the reference; handy for tests)."""
    for cls in (_qwen2_model_class(), _qwen2vl_model_class()):
        if cls is not None and "_sttm_original_forward" in cls.__dict__:
            cls.forward = cls._sttm_original_forward
            del cls._sttm_original_forward


def replace_qwen2_by_sparse_attn(pattern_name, **kwargs):
    if pattern_name == "quadtree":
        replace_qwen2_with_quadtree_attn(**kwargs)
        replace_qwen2vl_with_quadtree_attn(**kwargs)
    elif pattern_name == "tome":
        replace_qwen2_with_tome_attn(**kwargs)
        replace_qwen2vl_with_tome_attn(**kwargs)
    elif pattern_name == "quadtree-abl-pos":
        replace_qwen2_with_quadtree_attn_for_abl_pos(**kwargs)
    elif pattern_name == "octree":
        replace_qwen2_with_octree_attn(**kwargs)
    elif pattern_name == "pyrd":
        replace_qwen2_with_pyrd_attn(**kwargs)
    elif pattern_name == "dycoke-stage1":
        replace_qwen2_with_dycoke_stage1_attn(**kwargs)
        replace_qwen2vl_with_dycoke_stage1_attn(**kwargs)
    elif pattern_name in _UNIMPLEMENTED:
        raise NotImplementedError(f"{pattern_name} ({_UNIMPLEMENTED[pattern_name]}) is outside the MI355X hot-path build")
    else:
        raise NotImplementedError(f"{pattern_name} is not yet implemented")
