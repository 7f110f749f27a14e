"""DyCoke stage-1 temporal token pruning on the device: mirror of the reference's `dycoke_ttm`
(token_merging_utils/dycoke_merger.py:8-83), the function its "dycoke-stage1" hook calls
(token_merging_monkey_patch/dycoke_stage1_attn_monkey_patch.py:97).

Device only (HIP kernels in csrc/dycoke.hip), float32 / bfloat16 / float16; every size is known in advance, so nothing here
synchronises the stream.  Where two tokens of a frame have exactly the same similarity, `torch.topk` leaves their order
unspecified; this implementation puts the smaller token id first.
"""
import torch

from . import _lib

_ws_cache = _lib.BoundedCache(8)      # (device, stream) -> scratch; bounded, see _lib.BoundedCache


def dycoke_ttm(image_feature, num_frames, prune_ratio=0.7):
    """image_feature: [num_frames * P, C] float32 CUDA tensor.  Returns (combined_tokens [N', C], combined_indices [N'] int64)."""
    if not image_feature.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; there is no CPU fallback")
    if image_feature.dim() != 2:
        raise ValueError("expected a [num_frames * tokens_per_frame, C] tensor")
    codes = {torch.float32: _lib.STTM_F32, torch.bfloat16: _lib.STTM_BF16, torch.float16: _lib.STTM_F16}
    if image_feature.dtype not in codes:
        raise NotImplementedError(f"dtype {image_feature.dtype} is not supported (float32, bfloat16, float16)")
    T = int(num_frames)
    P = image_feature.shape[0] // T                                      # :11
    C = image_feature.shape[1]
    keep_ratio = 1 - prune_ratio                                         # :12
    k = int(keep_ratio * P)                                              # :37
    if T < 5 or P < 1:
        # the reference stacks an empty list of similarities for short clips (:24 / :63)
        raise RuntimeError("stack expects a non-empty TensorList")
    lib = _lib.load()
    x = image_feature.contiguous()
    dev = x.device
    rows = lib.sttm_dycoke_out_rows(T, P, k)
    nbytes = lib.sttm_dycoke_workspace_bytes(T, P, k)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        key = (dev, stream.cuda_stream)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _ws_cache[key] = ws
        out = torch.empty((rows, C), dtype=x.dtype, device=dev)
        idx = torch.empty(rows, dtype=torch.int64, device=dev)
        rc = lib.sttm_dycoke_ttm(x.data_ptr(), T, P, C, codes[x.dtype], k, ws.data_ptr(), ws.numel(), out.data_ptr(), idx.data_ptr(),
                                 stream.cuda_stream)
    _lib.raise_for(rc)
    return out, idx
