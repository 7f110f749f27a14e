"""A/B of the 256-tile ToMe match kernels' work split (PRODUCT build): tome_flat 0 (the j tiles of one a-tile in jsplit parts) against
2 (all tile products spread evenly over one workgroup per CU) and 1 (the default: whichever has the shorter per-CU critical path),
alternating call by call; outputs compared bit for bit first.  Run under rocprofv3 --kernel-trace --stats and read one size with
tools/prof_range.py (the kernel name is the same: the modes are told apart by their grid size in the trace, or run one mode per
process with MODES=0 / MODES=2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features, _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
MODES = tuple(int(m) for m in os.environ.get("MODES", "0,2,1").split(","))
for T in (180, 128, 100, 40):
    x32 = synth_video(T, 1024, 14, 14, seed=3, device=dev, gen_device=dev)
    for x in (x32, x32.to(torch.bfloat16)):
        for ratio in (0.5, 0.85):
            outs = []
            for m in MODES:
                lib.sttm_configure(b"tome_flat", m)
                outs.append(get_tome_features(x, ratio, "video"))
            same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
            print(f"T={T} {x.dtype} ratio {ratio}: tome_flat {MODES} bit-identical: {same}", flush=True)
        for m in MODES:
            lib.sttm_configure(b"tome_flat", m)
            get_tome_features(x, 0.5, "video"); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                get_tome_features(x, 0.5, "video")
            torch.cuda.synchronize()
            print(f"T={T} {x.dtype} tome_flat={m}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call (ratio 0.5)", flush=True)
lib.sttm_configure(b"tome_flat", 1)
