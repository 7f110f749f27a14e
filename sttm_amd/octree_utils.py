"""Octree baseline on the device: mirror of the reference's `get_octree_features`
(token_merging_utils/octree_utils.py:293-389), the function its "octree" hook calls
(token_merging_monkey_patch/octree_attn_monkey_patch.py:98).

The clip is cut into cubes of `W` frames (HIP kernels in csrc/octree.hip); the frames that do not fill a cube -- and clips
shorter than one cube -- go through the spatial quadtree per frame, exactly like the reference (:305-306, :375-378).
Device only.  The merged-token count is data dependent: one host read of a device counter per call.
"""
import torch

from . import _lib
from .quadtree_interface import get_quadtree_features

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_ws_cache = _lib.BoundedCache(8)      # (device, stream) -> scratch; bounded, see _lib.BoundedCache


def get_octree_features(_video_feature, threshold, root_level=0):
    """_video_feature: logical [T, C, H, W] CUDA tensor (channels-last views run zero-copy).  Returns features [N', C]."""
    if not _video_feature.is_cuda:
        raise RuntimeError("sttm_amd runs on the GPU only: the input must be a CUDA (ROCm) tensor; there is no CPU fallback")
    if _video_feature.dim() != 4:
        raise ValueError("expected a [T, C, H, W] tensor")
    if _video_feature.dtype not in _DTYPE_CODE:
        raise NotImplementedError(f"dtype {_video_feature.dtype} is not supported (float32, bfloat16, float16)")
    T, C, H, W = _video_feature.shape
    side = W                                                             # :296-297
    n_cube = T // side
    if n_cube == 0:                                                      # :305-306
        return get_quadtree_features(_video_feature, threshold, -1.0, root_level)[0]
    if H != W:
        raise NotImplementedError("the octree needs square frames (the reference's cube side is W)")
    drop = T % side
    body = _video_feature[:T - drop] if drop else _video_feature
    x = body.permute(0, 2, 3, 1)                                         # [T', H, W, C]
    if not x.is_contiguous() or x.data_ptr() % 16:
        x = x.contiguous()
    lib = _lib.load()
    dev = x.device
    code = _DTYPE_CODE[x.dtype]
    nbytes = lib.sttm_octree_workspace_bytes(n_cube, side, C, code, int(root_level))
    if nbytes == 0:
        # the reference indexes size_per_level[root_level] (:312-319): IndexError for an out-of-range level
        raise IndexError("list index out of range")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        key = (dev, stream.cuda_stream)
        cached = _ws_cache.get(key)
        if cached is None or cached[0].numel() < nbytes:
            cached = (torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
            _ws_cache[key] = cached
        ws, count = cached
        out = torch.empty((n_cube * side * side * side, C), dtype=x.dtype, device=dev)
        rc = lib.sttm_octree_build(x.data_ptr(), n_cube, side, C, code, float(threshold), int(root_level), ws.data_ptr(),
                                   ws.numel(), out.data_ptr(), count.data_ptr(), stream.cuda_stream)
    _lib.raise_for(rc)
    rem = get_quadtree_features(_video_feature[T - drop:], threshold, -1.0, root_level)[0] if drop else None      # :375-378
    n = int(count.item())
    return torch.cat([out[:n], rem], dim=0) if rem is not None else out[:n]
