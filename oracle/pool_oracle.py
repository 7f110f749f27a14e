"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's 2-D pooling step between the projector and the LLM
(`LlavaMetaForCausalLM.get_2dPool`, llava/model/llava_arch.py:173-198).

Parity pinned: checked against tests/golden/pool_*.npz, which tests/golden/make_golden_pool.py produced by running the
reference's own method (fp32 within 2e-6 absolute on unit-variance inputs -- the ATen kernels may contract multiplies and
adds differently; bf16 within one bf16 ulp; max pooling exact).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(sttm_amd/) never does.

The arithmetic is written out with numpy (no call into torch's pooling / interpolation operators):

  bilinear (:189-192, `F.interpolate(size=ceil(side / stride), mode="bilinear")`, align_corners=False):
      scale = in / out (float32);  src = max(fma(scale, dst + 0.5, -0.5), 0);  i0 = floor(src);  i1 = min(i0 + 1, in - 1)
      l1 = src - i0;  l0 = 1 - l1        (ATen: area_pixel_compute_source_index, UpSampleBilinear2d)
      out = h0 * (w0 * x[i0, j0] + w1 * x[i0, j1]) + h1 * (w0 * x[i1, j0] + w1 * x[i1, j1])      in float32
  average (:185-186, `F.avg_pool2d(stride)`): windows of stride x stride, floor(side / stride) outputs per axis, sum in
      float32 in row-major order, divided by stride^2
  max (:187-188, `F.max_pool2d(stride)`): same windows, maximum
  stride == 1 (:174-175): the input is returned as is.
"""
import math

import numpy as np
import torch


def pooled_side(side, stride, mode):
    if stride == 1:
        return side
    return math.ceil(side / stride) if mode == "bilinear" else side // stride


def _axis_taps(n_in, n_out):
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    # one rounding, like the fused multiply-add the ATen builds contract this expression to (the product of two float32
    # values is exact in float64); with two roundings index 8 of 27 -> 14 lands 1e-6 off (product >= 16, result < 16)
    src = (scale.astype(np.float64) * (dst + np.float32(0.5)).astype(np.float64) - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0))
    i0 = np.floor(src).astype(np.int64)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    l0 = (np.float32(1) - l1).astype(np.float32)
    return i0, i1, l0, l1


def get_2dpool(image_feature, stride=2, height=None, width=None, mode="bilinear"):
    """image_feature: [T, height*width, C] torch tensor (float32 / bfloat16 / float16).  Returns [T, oh*ow, C], same dtype."""
    if stride == 1:
        return image_feature
    T, n_tok, C = image_feature.shape
    if height is None:
        height = width = int(round(math.sqrt(n_tok)))
    if height * width != n_tok:
        raise RuntimeError("shape '[%d, %d, %d, -1]' is invalid for input of size %d" % (T, height, width, image_feature.numel()))
    x = image_feature.to(torch.float32).numpy().reshape(T, height, width, C)
    if mode == "bilinear":
        oh, ow = math.ceil(height / stride), math.ceil(width / stride)
        y0, y1, h0, h1 = _axis_taps(height, oh)
        x0, x1, w0, w1 = _axis_taps(width, ow)
        w0 = w0[None, None, :, None]; w1 = w1[None, None, :, None]
        top = w0 * x[:, y0][:, :, x0] + w1 * x[:, y0][:, :, x1]
        bot = w0 * x[:, y1][:, :, x0] + w1 * x[:, y1][:, :, x1]
        out = h0[None, :, None, None] * top + h1[None, :, None, None] * bot
    elif mode in ("average", "max"):
        oh, ow = height // stride, width // stride
        win = x[:, :oh * stride, :ow * stride].reshape(T, oh, stride, ow, stride, C)
        if mode == "max":
            out = win.max(axis=(2, 4))
        else:
            acc = np.zeros((T, oh, ow, C), dtype=np.float32)
            for dy in range(stride):
                for dx in range(stride):
                    acc = acc + win[:, :, dy, :, dx]
            out = acc / np.float32(stride * stride)
    else:
        raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")
    out = torch.from_numpy(np.ascontiguousarray(out.astype(np.float32))).to(image_feature.dtype)
    return out.reshape(T, oh * ow, C)


def resize_nearest(tokens, height, width, size):
    """The "pyrd" baseline's F.interpolate(video, size=size) (default mode "nearest",
    token_merging_monkey_patch/pyrd_attn_monkey_patch.py:99-102) on [T, height*width, C] tokens:
    src = min(floor(dst * (in / out)), in - 1) with a float32 scale and product (ATen nearest_neighbor_compute_source_index)."""
    T, n_tok, C = tokens.shape
    oh, ow = int(size[0]), int(size[1])

    def src(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        i = np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64)
        return np.minimum(i, n_in - 1)
    ys, xs = src(height, oh), src(width, ow)
    rows = torch.from_numpy((ys[:, None] * width + xs[None, :]).reshape(-1))
    return tokens[:, rows, :]
