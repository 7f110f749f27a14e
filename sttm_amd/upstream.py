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
