"""Where the host time of one drop-in call goes (MI355X box): stamps around the C call, the wait for N' and the return, plus the
same loop with the device idle in between (pure host cost).  usage: python tools/host_path.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib, get_quadtree_features
from sttm_amd import quadtree_interface as qi
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
pool = [synth_video(128, 1024, 14, 14, seed=i, device=dev, gen_device=dev) for i in range(8)]
for x in pool:
    get_quadtree_features(x, 0.85, 0.55, 1)
torch.cuda.synchronize()
lib = _lib.load()
stamps = {"launch": 0.0, "wait": 0.0}
orig_launch, orig_wait = lib.sttm_quadtree_merge_packed, lib.sttm_wait_counts_early
pc = time.perf_counter



class L:
    def __getattr__(self, k):
        return getattr(lib, k)

    def sttm_quadtree_merge_packed(self, *a):
        t0 = pc(); r = orig_launch(*a); stamps["launch"] += pc() - t0
        return r

    def sttm_wait_counts_early(self, *a):
        t0 = pc(); r = orig_wait(*a); stamps["wait"] += pc() - t0
        return r


n = 2000
torch.cuda.synchronize(); t0 = pc()
for i in range(n):
    get_quadtree_features(pool[i % 8], 0.85, 0.55, 1)
torch.cuda.synchronize(); base = (pc() - t0) / n
qi._lib = type("M", (), {k: getattr(_lib, k) for k in dir(_lib)})
qi._lib.load = staticmethod(lambda: L())
torch.cuda.synchronize(); t0 = pc()
for i in range(n):
    get_quadtree_features(pool[i % 8], 0.85, 0.55, 1)
torch.cuda.synchronize(); tot = (pc() - t0) / n
print(f"per call: uninstrumented {base * 1e6:.1f} us; instrumented {tot * 1e6:.1f} us = C call {stamps['launch'] / n * 1e6:.1f} + wait for N' {stamps['wait'] / n * 1e6:.1f} "
      f"+ other python {(tot - (stamps['launch'] + stamps['wait']) / n) * 1e6:.1f}")
# pure host cost: the same call with the device drained before every call (the wait then covers the whole pipeline)
stamps["launch"] = stamps["wait"] = 0.0
t_host = 0.0
for i in range(500):
    torch.cuda.synchronize()
    t0 = pc(); get_quadtree_features(pool[i % 8], 0.85, 0.55, 1); t_host += pc() - t0
print(f"device idle before each call: {t_host / 500 * 1e6:.1f} us per call = C call {stamps['launch'] / 500 * 1e6:.1f} + wait {stamps['wait'] / 500 * 1e6:.1f} + other python "
      f"{(t_host - stamps['launch'] - stamps['wait']) / 500 * 1e6:.1f}")
