"""Helpers to load the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype == "bfloat16":
        t = t.view(torch.bfloat16)
    elif dtype == "float16":
        t = t.view(torch.float16)
    return t


def load_case(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    dtype = meta.get("dtype", "float32")
    case = dict(meta=meta, name=meta["name"])
    x = _t(z["x_thwc"], dtype)                   # [T, H, W, C] memory
    case["x"] = x.permute(0, 3, 1, 2)            # production layout: [T, C, H, W] view
    case["feat"] = _t(z["feat"], dtype)
    if meta["fn"] == "quadtree":
        case["npatch"] = torch.from_numpy(z["npatch"])
        case["tlbr"] = torch.from_numpy(z["tlbr"])
        if "pos_cos_thwc" in z:
            case["pos_embs"] = (_t(z["pos_cos_thwc"], dtype).permute(0, 3, 1, 2), _t(z["pos_sin_thwc"], dtype).permute(0, 3, 1, 2))
            case["out_pos"] = (_t(z["out_cos"], dtype), _t(z["out_sin"], dtype))
    else:
        case["idx"] = torch.from_numpy(z["idx"])
    return case


def case_paths(prefixes):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        if os.path.basename(p).startswith(tuple(prefixes)):
            out.append(p)
    return out


def kat():
    with open(os.path.join(GOLDEN_DIR, "kat.json")) as f:
        return json.load(f)


def quadtree_kwargs(meta):
    kw = dict(meta["kw"])
    thr = kw.pop("threshold")
    kw.pop("pos", None)            # channel count of the position embeddings stored in the fixture
    return thr, kw


# ---- upstream pooling vectors (tests/golden/pool_*.npz, made by make_golden_pool.py) -------------------------
POOL_GOLDEN = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pool_*.npz")))


def load_pool_case(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    return meta, _t(z["x"], meta["dtype"]), _t(z["y"], meta["dtype"])


def pool_close_enough(out, exp, meta):
    """max pooling / stride 1: exact; bf16: one ulp; fp32: 2e-6 absolute on unit-variance inputs (the ATen kernels may
    contract multiplies and adds differently from the one-rounding-per-operation restatement)."""
    assert out.shape == exp.shape and out.dtype == exp.dtype
    if meta["mode"] == "max" or meta["stride"] == 1:
        return torch.equal(out, exp)
    if exp.dtype == torch.bfloat16:
        d = (out.view(torch.int16).int() - exp.view(torch.int16).int()).abs()
        return int(d.max()) <= 1
    return float((out - exp).abs().max()) <= 2e-6


# ---- DyCoke stage-1 vectors (tests/golden/dyc_*.npz, made by make_golden_dycoke.py) --------------------------
DYCOKE_GOLDEN = sorted(glob.glob(os.path.join(GOLDEN_DIR, "dyc_*.npz")))


def load_dycoke_case(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    return meta, torch.from_numpy(z["x"]), torch.from_numpy(z["feat"]), torch.from_numpy(z["idx"])


# ---- octree vectors (tests/golden/oct_*.npz, made by make_golden_octree.py) ----------------------------------
OCTREE_GOLDEN = sorted(glob.glob(os.path.join(GOLDEN_DIR, "oct_*.npz")))


def load_octree_case(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    x = _t(z["x_thwc"], meta["dtype"]).permute(0, 3, 1, 2)          # [T, C, H, W] view of channels-last memory
    return meta, x, _t(z["feat"], meta["dtype"])
