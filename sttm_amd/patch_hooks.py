"""Tensor glue of the reference's patched decoder forward, as plain functions.

These restate the ~25 lines the reference inserts before decoder layer `sa_start_layer_idx` during prefill:
    LLaVA-Video / OneVision  token_merging_monkey_patch/quadtree_attn_monkey_patch.py:88-117
    Qwen2-VL                 token_merging_qwen2vl_monkey_patch/quadtree_attn_monkey_patch.py:88-115
    ToMe                     token_merging_monkey_patch/tome_attn_monkey_patch.py:88-107
The visual slice is handed to the merge function as a [T, C, H, W] *view* of [T, H, W, C] memory (no copy),
exactly like einops.rearrange(visual[0], "(T H W) C -> T C H W") in the reference.
"""
import math

import torch


def _video_view(visual, T, H, W):
    C = visual.shape[-1]
    return visual.reshape(T, H, W, C).permute(0, 3, 1, 2)


def split_prompt(hidden_states, start, length):
    end = start + length
    return hidden_states[:, :start], hidden_states[:, start:end], hidden_states[:, end:]


def _merge_concat(hidden_states, start, length, video, merge_fn, merge_into_fn, args, kwargs):
    """system ++ merge(video) ++ instruction  (quadtree_attn_monkey_patch.py:101-105).

    With `merge_into_fn` (sttm_amd.get_quadtree_features_into) the merged rows are written by the kernels straight into
    the new hidden-state buffer: no intermediate [N', C] tensor and no copy of it by torch.cat.
    The new buffer has the OLD sequence length (N' <= length), the result is its leading [:, :S'] view."""
    sys_f, _, inst_f = split_prompt(hidden_states, start, length)
    if merge_into_fn is None or hidden_states.size(0) != 1:
        feat, npatch, tlbr = merge_fn(video, *args, **kwargs)
        return torch.cat([sys_f, feat.unsqueeze(0), inst_f], dim=1), tlbr, None
    new = torch.empty_like(hidden_states, memory_format=torch.contiguous_format)
    new[:, :start].copy_(sys_f)
    # the kernels also write merged_token_1d_idx (t*H*W + y1*W + x1, :103-104) when the merge function offers it: three
    # elementwise launches of the caller less
    idx = None
    if getattr(merge_into_fn, "returns_idx", False):
        feat, npatch, tlbr, idx = merge_into_fn(new[0, start:], video, *args, return_idx=True, **kwargs)
    else:
        feat, npatch, tlbr = merge_into_fn(new[0, start:], video, *args, **kwargs)
    n = feat.size(0)
    new[:, start + n:start + n + inst_f.size(1)].copy_(inst_f)
    return new[:, :start + n + inst_f.size(1)], tlbr, idx


def quadtree_merge_llava(hidden_states, position_ids, start, length, T, merge_fn, threshold, temporal_thresh,
                         root_level, weighted_avg, slow_ver=False, head_dim=None, merge_into_fn=None):
    """Returns (merged hidden_states [1, S', C], position_ids[:, :S'], merged_token_1d_idx)."""
    H = W = int(math.sqrt(length // T))                                   # :97 (needs mm_newline_position=no_token)
    video = _video_view(hidden_states[0, start:start + length], T, H, W)
    merged, tlbr, idx = _merge_concat(hidden_states, start, length, video, merge_fn, merge_into_fn,
                                      (threshold, temporal_thresh, root_level, weighted_avg),
                                      dict(slow_ver=slow_ver, head_dim=head_dim))
    if idx is None:
        idx = tlbr[:, 0] * (H * W) + tlbr[:, 1] * W + tlbr[:, 2]          # :103-104
    return merged, position_ids[:, :merged.size(1)], idx                   # :114 (truncate, not gather)


def quadtree_merge_qwen2vl(hidden_states, position_ids, start, length, T, H, W, merge_fn, threshold, temporal_thresh,
                           root_level, weighted_avg, slow_ver=False, merge_into_fn=None):
    """position_ids is the 3-D mRoPE tensor [3, B, S]; the visual part is GATHERED by the merged index (:109-113).
    Returns (merged hidden_states, position_ids, cache_position, merged_token_1d_idx)."""
    end = start + length
    video = _video_view(hidden_states[0, start:end], T, H, W)
    merged, tlbr, idx = _merge_concat(hidden_states, start, length, video, merge_fn, merge_into_fn,
                                      (threshold, temporal_thresh, root_level, weighted_avg), dict(slow_ver=slow_ver))
    if idx is None:
        idx = tlbr[:, 0] * (H * W) + tlbr[:, 1] * W + tlbr[:, 2]
    vis_pos = position_ids[:, :, start:end][:, :, idx.long()]
    pos = torch.cat([position_ids[:, :, :start], vis_pos, position_ids[:, :, end:]], dim=-1)
    cache_position = torch.arange(merged.size(1), device=merged.device, dtype=torch.int)      # :114
    return merged, pos, cache_position, idx


def tome_merge(hidden_states, position_ids, start, length, T, merge_fn, prune_ratio, tome_ver, H=None, W=None,
               gather_positions=False):
    """ToMe hook.  LLaVA (token_merging_monkey_patch/tome_attn_monkey_patch.py:88-107): H = W = sqrt(len / T), position_ids
    [B, S] are TRUNCATED to the new length (:105).  Qwen2-VL (token_merging_qwen2vl_monkey_patch/tome_attn_monkey_patch.py:
    88-111, gather_positions=True): H, W come from the module, and the 3-D mRoPE ids [3, B, S] of the visual part are
    GATHERED by the ToMe token index, system / instruction ids kept (:105-108)."""
    sys_f, vis_f, inst_f = split_prompt(hidden_states, start, length)
    end = start + length
    if H is None:
        H = int(math.sqrt(length // T))
        W = (length // T) // H
    video = _video_view(vis_f[0], T, H, W)
    feat, token_idx = merge_fn(video, prune_ratio, tome_ver)
    merged = torch.cat([sys_f, feat.unsqueeze(0), inst_f], dim=1)
    if gather_positions:
        vis_pos = position_ids[:, :, start:end][:, :, token_idx]
        pos = torch.cat([position_ids[:, :, :start], vis_pos, position_ids[:, :, end:]], dim=-1)
    else:
        pos = position_ids[:, :merged.size(1)]
    return merged, pos, token_idx


def pyrd_resize(hidden_states, position_ids, start, length, T, tgt_size, resize_fn):
    """The "pyrd" baseline (pyrd_attn_monkey_patch.py:88-112): every frame of the visual slice is resized to
    tgt_size x tgt_size with F.interpolate's default (nearest) mode.  resize_fn(tokens [T, H*W, C], H, W, (s, s)).
    Returns (hidden_states, position_ids[:, :S'], new visual length)."""
    sys_f, vis_f, inst_f = split_prompt(hidden_states, start, length)
    H = int(math.sqrt(length // T))                                       # :97
    C = vis_f.shape[-1]
    resized = resize_fn(vis_f[0].reshape(T, H * H, C), H, H, (tgt_size, tgt_size))
    resized = resized.reshape(1, T * tgt_size * tgt_size, C)
    merged = torch.cat([sys_f, resized, inst_f], dim=1)                   # :103
    return merged, position_ids[:, :merged.size(1)], resized.size(1)       # :101, :108


def dycoke_merge(hidden_states, position_ids, start, length, T, ttm_fn, prune_ratio, gather_positions=False):
    """DyCoke stage-1 hook: LLaVA (dycoke_stage1_attn_monkey_patch.py:88-107) truncates position_ids; Qwen2-VL
    (token_merging_qwen2vl_monkey_patch/dycoke_stage1_attn_monkey_patch.py:88-108, gather_positions=True) gathers the 3-D
    mRoPE ids of the kept tokens.  ttm_fn(tokens [T*P, C], T, prune_ratio) -> (tokens, flat ids)."""
    sys_f, vis_f, inst_f = split_prompt(hidden_states, start, length)
    end = start + length
    feat, idx = ttm_fn(vis_f[0], T, prune_ratio)
    merged = torch.cat([sys_f, feat.unsqueeze(0), inst_f], dim=1)
    if gather_positions:
        vis_pos = position_ids[:, :, start:end][:, :, idx]
        pos = torch.cat([position_ids[:, :, :start], vis_pos, position_ids[:, :, end:]], dim=-1)
    else:
        pos = position_ids[:, :merged.size(1)]
    return merged, pos, idx


def octree_merge(hidden_states, position_ids, start, length, T, octree_fn, threshold, root_level):
    """Octree hook (octree_attn_monkey_patch.py:87-106): features only, position_ids truncated."""
    sys_f, vis_f, inst_f = split_prompt(hidden_states, start, length)
    H = int(math.sqrt(length // T))                                       # :96
    video = _video_view(vis_f[0], T, H, H)
    feat = octree_fn(video, threshold, root_level)
    merged = torch.cat([sys_f, feat.unsqueeze(0), inst_f], dim=1)
    return merged, position_ids[:, :merged.size(1)]


def quadtree_merge_abl_pos(hidden_states, position_ids, position_embeddings, start, length, T, merge_fn, threshold,
                           temporal_thresh, root_level, weighted_avg, pos_emb_ver, pos_emb_weighted_avg, rotary_fn,
                           slow_ver=False, head_dim=None):
    """Position-embedding ablation hook (quadtree_attn_monkey_patch_for_abl_pos.py:88-136).
    pos_emb_ver 0: re-number the shorter sequence (the standard hook); 1: merge the RoPE (cos, sin) rows alongside the features
    (`pos_embs=`); 2: keep every merged token's ORIGINAL position id (gather by the merged index).
    rotary_fn(hidden_states, position_ids) -> (cos, sin).  Returns (hidden_states, position_ids, position_embeddings, idx)."""
    sys_f, vis_f, inst_f = split_prompt(hidden_states, start, length)
    end = start + length
    H = W = int(math.sqrt(length // T))                                   # :97
    video = _video_view(vis_f[0], T, H, W)
    pos_video = None
    if pos_emb_ver > 0:                                                   # :100-104
        pos_video = tuple(_video_view(p[0, start:end], T, H, W) for p in position_embeddings)
    res = merge_fn(video, threshold, temporal_thresh, root_level, weighted_avg, slow_ver=slow_ver, head_dim=head_dim,
                   pos_embs=pos_video, pos_emb_weighted_avg=pos_emb_weighted_avg)
    if pos_emb_ver > 0:
        feat, _, tlbr, merged_pos = res
    else:
        feat, _, tlbr = res
    idx = tlbr[:, 0] * (H * W) + tlbr[:, 1] * W + tlbr[:, 2]              # :113-114
    merged = torch.cat([sys_f, feat.unsqueeze(0), inst_f], dim=1)
    n = merged.size(1)
    if pos_emb_ver == 0:                                                  # :120-123
        position_ids = position_ids[:, :n]
        position_embeddings = rotary_fn(merged, position_ids)
    elif pos_emb_ver == 1:                                                # :124-128
        position_embeddings = tuple(torch.cat([p[:, :start], mp.unsqueeze(0), p[:, end:]], dim=1)
                                    for p, mp in zip(position_embeddings, merged_pos))
        position_ids = position_ids[:, :n]
    elif pos_emb_ver == 2:                                                # :129-135
        vis_ids = position_ids[:, start:end][:, idx.long()]
        position_ids = torch.cat([position_ids[:, :start], vis_ids, position_ids[:, end:]], dim=-1)
        position_embeddings = rotary_fn(merged, position_ids)
    return merged, position_ids, position_embeddings, idx
