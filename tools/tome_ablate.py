"""ToMe 256-tile match kernel: where the time goes (DEVELOPMENT build, `python -m sttm_amd.build --dev`; run under rocprofv3 or
alone).  STTM_TOME_ABL is read per launch: 0 = the kernel, 1 = no DMA after the prologue (MFMAs + fragment reads + barriers on
stale LDS), 2 = no MFMAs (DMA + barriers + fragment reads), 3 = the kernel with the next step's fragment reads issued AFTER the first
eight MFMAs of a step (valid outputs, compared with mode 0 here).  Outputs of modes 1 / 2 are invalid; only the time of the FIRST match
of a step (12544 x 12544 x 1024 at T = 128) is read, with HIP events around the whole step minus the other kernels' share being
irrelevant here -- so the step is cut to one iteration (ratio 0.5) and the same non-match kernels run in every mode."""
import os, sys, time
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_tome_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
T = int(os.environ.get("T", "128"))
x = synth_video(T, 1024, 14, 14, seed=3, device=dev, gen_device=dev)
os.environ["STTM_TOME_ABL"] = "0"
f0, i0 = get_tome_features(x, 0.5, "video")
os.environ["STTM_TOME_ABL"] = "3"
f3, i3 = get_tome_features(x, 0.5, "video")
print("mode 3 == mode 0:", bool(torch.equal(f0, f3) and torch.equal(i0, i3)))
for rep in range(2):
    for mode in (0, 1, 2, 3):
        os.environ["STTM_TOME_ABL"] = str(mode)
        get_tome_features(x, 0.5, "video")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            get_tome_features(x, 0.5, "video")
        torch.cuda.synchronize()
        print(f"STTM_TOME_ABL={mode}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per get_tome_features call (T={T}, ratio 0.5)")

xb = x.to(torch.bfloat16)
os.environ["STTM_TOME_ABL"] = "0"
f0, i0 = get_tome_features(xb, 0.5, "video")
os.environ["STTM_TOME_ABL"] = "3"
f3, i3 = get_tome_features(xb, 0.5, "video")
print("bf16: mode 3 == mode 0:", bool(torch.equal(f0, f3) and torch.equal(i0, i3)))
for rep in range(2):
    for mode in (0, 3):
        os.environ["STTM_TOME_ABL"] = str(mode)
        get_tome_features(xb, 0.5, "video")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            get_tome_features(xb, 0.5, "video")
        torch.cuda.synchronize()
        print(f"bf16 STTM_TOME_ABL={mode}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per get_tome_features call (T={T}, ratio 0.5)")
