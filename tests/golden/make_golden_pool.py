#!/usr/bin/env python3
"""Golden vectors for the upstream 2-D pooling step, made by RUNNING THE REFERENCE's
`LlavaMetaForCausalLM.get_2dPool` (llava/model/llava_arch.py:173-198) in the build container.

The reference package cannot be imported as a package here (its model zoo needs an older transformers), so this script
compiles that ONE method out of the reference file at generation time (ast -> exec, nothing is copied into the repo) and
calls it with a stand-in `self` that only carries `config.mm_spatial_pool_mode` and `num_patches_per_side`.
Only data is written: inputs, parameters, expected outputs.

    python tests/golden/make_golden_pool.py        # rewrites tests/golden/pool_*.npz
"""
import ast
import json
import math
import os
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("STTM_REFERENCE", "/root/reference")


def reference_get_2dpool():
    path = os.path.join(REF, "llava", "model", "llava_arch.py")
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "get_2dPool":
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"math": math, "nn": nn, "torch": torch}
            exec(compile(mod, path, "exec"), ns)
            return ns["get_2dPool"]
    raise RuntimeError("get_2dPool not found in the reference")


def to_np(t):
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy()
    return t.contiguous().numpy()


CASES = [
    dict(name="pool_bilinear_27_s2", mode="bilinear", T=3, side=27, C=32, stride=2, dtype="float32", seed=1),
    dict(name="pool_bilinear_27_s2_bf16", mode="bilinear", T=2, side=27, C=64, stride=2, dtype="bfloat16", seed=2),
    dict(name="pool_bilinear_27_s3", mode="bilinear", T=2, side=27, C=16, stride=3, dtype="float32", seed=3),
    dict(name="pool_bilinear_24_s2_width", mode="bilinear", T=2, side=24, C=16, stride=2, dtype="float32", seed=4, width=24),
    dict(name="pool_average_27_s2", mode="average", T=3, side=27, C=32, stride=2, dtype="float32", seed=5),
    dict(name="pool_average_24_s2_bf16", mode="average", T=2, side=24, C=64, stride=2, dtype="bfloat16", seed=6, width=24),
    dict(name="pool_max_27_s2", mode="max", T=2, side=27, C=32, stride=2, dtype="float32", seed=7),
    dict(name="pool_max_27_s4_bf16", mode="max", T=2, side=27, C=64, stride=4, dtype="bfloat16", seed=8),
    dict(name="pool_stride1", mode="bilinear", T=2, side=9, C=8, stride=1, dtype="float32", seed=9, width=9),
]


def main():
    fn = reference_get_2dpool()
    for c in CASES:
        g = torch.Generator().manual_seed(c["seed"])
        dtype = getattr(torch, c["dtype"])
        x = torch.randn(c["T"], c["side"] * c["side"], c["C"], generator=g).to(dtype)
        me = types.SimpleNamespace(config=types.SimpleNamespace(mm_spatial_pool_mode=c["mode"]),
                                   get_vision_tower=lambda side=c["side"]: types.SimpleNamespace(num_patches_per_side=side))
        y = fn(me, x, stride=c["stride"], width=c.get("width", -1))
        meta = dict(c, fn="pool2d", out_tokens=int(y.shape[1]))
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), meta=json.dumps(meta), x=to_np(x), y=to_np(y))
        print(c["name"], tuple(x.shape), "->", tuple(y.shape))


if __name__ == "__main__":
    main()
