"""The step right before the merge path: 2-D pooling of the projected vision tokens.

Mirror of `LlavaMetaForCausalLM.get_2dPool` (llava/model/llava_arch.py:173-198 of the reference) as a free function:
the reference reads `mm_spatial_pool_mode` from the model config and the side from the vision tower; here both are
arguments.  LLaVA-Video: [T, 729, C] (27 x 27 SigLIP patches after the projector) -> bilinear -> [T, 196, C].
Device only (HIP kernel `k_pool2d`); tokens stay channels-last, no NCHW round trip.
"""
import math

import torch

from . import _lib

_MODES = {"average": 0, "max": 1, "bilinear": 2}
_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def get_2dPool(image_feature, stride=2, width=-1, mode="bilinear", num_patches_per_side=None):
    """image_feature: [num_frames, height*width, C] CUDA tensor.  `width` == -1 takes the (square) side from
    `num_patches_per_side` (the reference asks the vision tower) or from sqrt(num_tokens).  Returns
    [num_frames, out_h*out_w, C] in the input dtype."""
    if stride == 1:                                                     # :174-175
        return image_feature
    if mode not in _MODES:
        raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")    # :194-195
    if not image_feature.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; there is no CPU fallback")
    if image_feature.dim() != 3:
        raise ValueError("expected a [num_frames, num_tokens, C] tensor")
    if image_feature.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {image_feature.dtype} is not supported (float32, bfloat16, float16)")
    T, n_tok, C = image_feature.shape
    if width == -1:
        height = width = num_patches_per_side if num_patches_per_side is not None else int(round(math.sqrt(n_tok)))
    else:
        height = width
    if height * width != n_tok:                                         # the reference's .view() fails the same way (:181)
        raise RuntimeError("shape '[%d, %d, %d, -1]' is invalid for input of size %d" % (T, height, width, image_feature.numel()))
    lib = _lib.load()
    x = image_feature.contiguous()
    oh = lib.sttm_pool2d_out_side(height, stride, _MODES[mode])
    ow = lib.sttm_pool2d_out_side(width, stride, _MODES[mode])
    if oh < 1 or ow < 1:
        _lib.raise_for(_lib.ERR_ARG)
    out = torch.empty((T, oh * ow, C), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.sttm_pool2d(x.data_ptr(), T, height, width, C, _DTYPE_CODE[x.dtype], _MODES[mode], int(stride),
                             out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    _lib.raise_for(rc)
    return out


def resize_nearest(tokens, height, width, size):
    """F.interpolate(video, size=size) with the default mode ("nearest"), on channels-last tokens:
    tokens [T, height*width, C] -> [T, size[0]*size[1], C] (the "pyrd" baseline, pyrd_attn_monkey_patch.py:99-102)."""
    if not tokens.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; there is no CPU fallback")
    if tokens.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {tokens.dtype} is not supported (float32, bfloat16, float16)")
    T, n_tok, C = tokens.shape
    if height * width != n_tok:
        raise RuntimeError("token count %d is not %d x %d" % (n_tok, height, width))
    oh, ow = int(size[0]), int(size[1])
    x = tokens.contiguous()
    out = torch.empty((T, oh * ow, C), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.sttm_resize_nearest(x.data_ptr(), T, height, width, C, _DTYPE_CODE[x.dtype], oh, ow, out.data_ptr(),
                                     torch.cuda.current_stream(x.device).cuda_stream)
    _lib.raise_for(rc)
    return out


# ---------------------------------------------------------------------------------------------------
# On-disk feature formats of the reference's pre-extraction scripts (second half)
# ---------------------------------------------------------------------------------------------------
def _load_pt(path):
    feat = torch.load(path, map_location="cpu", weights_only=True)
    if not torch.is_tensor(feat):
        raise ValueError(f"{path}: expected one tensor (torch.save(video_feats.cpu(), ...)), got {type(feat).__name__}")
    return feat


def load_llavavideo_features(path, device, projector=None, stride=2, mode="bilinear", dtype=torch.bfloat16):
    """`<vid>.pt` written by llava/eval/video_feat_llavavideo.py:89-95: ONE tensor [T, 729, 1152] bfloat16 -- the SigLIP tower's
    27 x 27 patch tokens per frame, BEFORE the multimodal projector.  The reference's evaluation moves it to the GPU
    (eval_vidqa_by_feat_llavavideo.py:213), projects it (llava_arch.py:238, `mm_projector`: model weights, the caller's) and pools
    27 x 27 -> 14 x 14 (`get_2dPool`, llava_arch.py:173-198).  This does the same with the HIP pooling kernel and returns the
    video in the layout the merge path takes: a logical [T, C, 14, 14] view of [T, 14, 14, C] memory (zero-copy for
    get_quadtree_features / get_tome_features), plus the pooled side.
    projector: callable [T, 729, Cv] -> [T, 729, C] on the device, or None when the file already holds projected tokens."""
    feat = _load_pt(path)
    if feat.dim() != 3:
        raise ValueError(f"{path}: expected [T, tokens, C], got {tuple(feat.shape)}")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("sttm_amd runs on the GPU only: load onto a CUDA (ROCm) device; there is no CPU fallback")
    x = feat.to(dtype).to(dev, non_blocking=True)
    if projector is not None:
        x = projector(x)
    T, n_tok, C = x.shape
    pooled = get_2dPool(x, stride=stride, mode=mode)                     # [T, side*side, C]
    side = int(round(math.sqrt(pooled.shape[1])))
    return pooled.reshape(T, side, side, C).permute(0, 3, 1, 2), side


def load_qwen2vl_features(path, device, dtype=torch.bfloat16):
    """`<vid>.pt` written by llava/eval/video_feat_qwen2vl.py:72-79: ONE tensor [T, H, W, C] (the merged-patch grid of the Qwen2-VL
    vision tower, variable H x W per video).  The evaluation casts to bfloat16 on the GPU (eval_vidqa_by_feat_qwen2vl.py:152) and
    flattens to (T H W) C; the hook later views the slice as [T, C, H, W].  Returns that view (zero-copy) and (T, H, W)."""
    feat = _load_pt(path)
    if feat.dim() != 4:
        raise ValueError(f"{path}: expected [T, H, W, C], got {tuple(feat.shape)}")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("sttm_amd runs on the GPU only: load onto a CUDA (ROCm) device; there is no CPU fallback")
    x = feat.to(dtype).to(dev, non_blocking=True).contiguous()
    T, H, W, _ = x.shape
    return x.permute(0, 3, 1, 2), (T, H, W)
