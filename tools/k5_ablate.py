"""Group-mean kernel with / without the member scan (DEVELOPMENT build; mode 1 gives invalid outputs, only the time is read)."""
import ctypes, os, sys
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.quadtree_interface import quadtree_merge_raw
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
T, H, W = int(os.environ.get("T", "128")), int(os.environ.get("H", "14")), int(os.environ.get("W", "14"))
THR, TTHR, RL = float(os.environ.get("THR", "0.85")), float(os.environ.get("TTHR", "0.55")), int(os.environ.get("RL", "1"))
pool = [synth_video(T, 1024, H, W, seed=i, device=dev, gen_device=dev) for i in range(8 if H * W < 400 else 4)]
ev = _lib.KernelEvents()
for mode in (0, 1, 0, 1):
    lib.sttm_dev_k5_mode(mode)
    tot = [0.0] * 4; n = 0
    for it in range(48):
        quadtree_merge_raw(pool[it % len(pool)], THR, TTHR, RL, False, None, events=ev)
        ms = ev.elapsed_ms()
        if it >= 8:
            tot = [a + b for a, b in zip(tot, ms)]; n += 1
    print(f"k5_mode={mode}: " + ", ".join(f"{k}={v / n * 1e3:.1f} us" for k, v in zip(_lib.KernelEvents.NAMES, tot)))
lib.sttm_dev_k5_mode(0)
