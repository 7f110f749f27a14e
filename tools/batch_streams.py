"""Batched extension: throughput against the number of side streams (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features_batch
from sttm_amd import quadtree_interface as QI
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
pool = [synth_video(128, 1024, 14, 14, seed=s, device=dev, gen_device=dev) for s in range(8)]
vids = [pool[i % 8] for i in range(32)]
for ns in (1, 2, 3, 4, 6, 8):
    for _ in range(3): get_quadtree_features_batch(vids, 0.85, 0.55, 1, n_streams=ns)
    torch.cuda.synchronize()
    reps = 10
    enq = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        del QI._BATCH_TIMING[:]; QI._BATCH_TIMING.append(0.0)
        ta = time.perf_counter()
        out = get_quadtree_features_batch(vids, 0.85, 0.55, 1, n_streams=ns)
        enq += QI._BATCH_TIMING[1] - ta
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del QI._BATCH_TIMING[:]
    print(f"n_streams={ns}: {reps * len(vids) / dt:.0f} videos/s ({dt / reps / len(vids) * 1e6:.1f} us per video; the host spends "
          f"{enq / reps / len(vids) * 1e6:.1f} us per video enqueueing)")
