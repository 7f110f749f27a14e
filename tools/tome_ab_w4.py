"""ToMe match: the four-wave form of the 256-tile kernel (tome_split = 7: 128 x 128 wave tiles, accumulators in literal AGPRs) against
the eight-wave form (tome_split = 6 / default), interleaved in one process on one box.  Whole get_tome_features calls at ratio 0.5
(one iteration: normalise + match + rank + merge), videos/s; outputs compared first.
    python tools/tome_ab_w4.py        (environment: T = 128 / 180, REPS)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib, get_tome_features
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
T, REPS, N = int(os.environ.get("T", "180")), int(os.environ.get("REPS", "5")), int(os.environ.get("N", "20"))
for dt in (torch.bfloat16, torch.float32, torch.float16):
    pool = [synth_video(T, 1024, 14, 14, seed=3 + i, device=dev, gen_device=dev).to(dt) for i in range(4)]
    modes = {"eight waves (split 6)": 6, "four waves (split 7)": 7}
    outs = {}
    for name, m in modes.items():
        _lib.configure(tome_split=m)
        outs[name] = get_tome_features(pool[0], 0.5, "video")
        torch.cuda.synchronize()
    (fa, ia), (fb, ib) = outs.values()
    same_ids = torch.equal(ia, ib)
    diff = float((fa.float() - fb.float()).abs().max()) if fa.shape == fb.shape else float("nan")
    res = {k: [] for k in modes}
    for rep in range(REPS):
        for name, m in modes.items():
            _lib.configure(tome_split=m)
            get_tome_features(pool[0], 0.5, "video"); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(N):
                get_tome_features(pool[i % 4], 0.5, "video")
            torch.cuda.synchronize()
            res[name].append(N / (time.perf_counter() - t0))
    _lib.configure(tome_split=2)
    print(f"T={T} {dt}: ids equal {same_ids}, max feature diff {diff:.2e}")
    for name in modes:
        r = sorted(res[name])
        print(f"   {name:28s} median {r[len(r) // 2]:8.1f}  max {r[-1]:8.1f} videos/s")
