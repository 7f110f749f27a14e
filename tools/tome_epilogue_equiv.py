"""DEVELOPMENT build (STTM_LIB=dev): the two-pass running max of the 256-tile ToMe match kernels (fp32 maximum, one rounding per row, first
candidate at or above the tie floor) against the one-pass sweep it replaced (STTM_TOME_ABL=8: round, compare, select per score) on inputs built to
hit the corners of the tie-floor arithmetic: all-negative scores (every b row opposes every a row), exact zeros and exact ones (one-hot rows:
R = +-0, R = 1), scores packed around 16-bit rounding boundaries (near-duplicate rows), NaN rows (zero tokens), tiny clips with partial tiles.
Best scores (bits) and argmax indices of sttm_tome_step must be identical for every dtype and both kernel forms.
    STTM_LIB=dev python tools/tome_epilogue_equiv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("STTM_LIB") == "dev", "needs the development build: STTM_LIB=dev (python -m sttm_amd.build --dev)"
import torch
from sttm_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)


def clip(kind, n, C):
    x = torch.randn(n, C, generator=g)
    if kind == "negative":                       # a rows near +d, b rows near -d: every score is negative
        d = torch.randn(C, generator=g)
        x = 0.3 * x
        x[0::2] += d
        x[1::2] -= d
    elif kind == "onehot":                       # scores are exactly 0 or 1 (and -1 for the flipped ones)
        x = torch.zeros(n, C)
        hot = torch.randint(0, C, (n,), generator=g)
        x[torch.arange(n), hot] = 1.0
        x[torch.randperm(n, generator=g)[: n // 7]] *= -1.0
    elif kind == "boundary":                     # near-duplicates: scores crowd the top few 16-bit values below 1
        base = torch.randn(16, C, generator=g)
        x = base[torch.randint(0, 16, (n,), generator=g)] + 0.02 * x
    elif kind == "nan":
        x[torch.randperm(n, generator=g)[: max(2, n // 50)]] = 0.0
    elif kind == "tiny":                         # opposing rows with small noise: negative scores of all magnitudes
        x = 1e-3 * x
        x[0::2, 0] += 1.0
        x[1::2, 0] -= torch.rand(n // 2, generator=g)
    return x


def step(x, code, split, abl):
    os.environ["STTM_TOME_ABL"] = str(abl)
    _lib.configure(tome_split=split)
    n, C = x.shape
    r, na = n // 2, (n + 1) // 2
    nbytes = lib.sttm_tome_workspace_bytes(n, C, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    xo = torch.empty((n - r, C), dtype=x.dtype, device=dev); so = torch.empty(n - r, device=dev); io = torch.empty(n - r, dtype=torch.int64, device=dev)
    nmax = torch.empty(na, device=dev); nidx = torch.empty(na, dtype=torch.int32, device=dev)
    rc = lib.sttm_tome_step(x.data_ptr(), None, None, n, C, 1, r, code, ws.data_ptr(), nbytes, xo.data_ptr(), so.data_ptr(), io.data_ptr(),
                            nmax.data_ptr(), nidx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.raise_for(rc)
    torch.cuda.synchronize()
    return nmax.view(torch.int32).cpu(), nidx.cpu(), io.cpu()


checked = 0
try:
    for dtype, code in ((torch.bfloat16, 1), (torch.float16, 2), (torch.float32, 0)):
        for kind in ("random", "negative", "onehot", "boundary", "nan", "tiny"):
            for n, C in ((300, 64), (1100, 256), (7001, 128), (9000, 1024)):
                x = clip(kind, n, C).to(dtype).to(dev)
                for split in ((4, 7) if code else (6, 7)):
                    a = step(x, code, split, 8)
                    b = step(x, code, split, 0)
                    for name, u, v in zip(("best scores", "argmax", "output ids"), a, b):
                        assert torch.equal(u, v), f"{dtype} {kind} n={n} C={C} tome_split={split}: {name} differ in {int((u != v).sum())} places"
                    checked += 1
finally:
    os.environ.pop("STTM_TOME_ABL", None)
    _lib.configure(tome_split=2)
print(f"two-pass running max == one-pass sweep on {checked} (dtype, input kind, size, kernel form) combinations: best-score bits, argmax and output ids identical")
