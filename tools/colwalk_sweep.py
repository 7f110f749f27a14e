"""Column-walk spatial stage (csrc/spatial_col.inc) against the spatial + pair kernels in the batch entry point: videos/s under
library switches, interleaved in ONE process on one box (same-box A/B).  Run on the GPU box:
    python tools/colwalk_sweep.py ["key=v,key=v" ...]        (default: a built-in list)
Environment: T, C, DT (f32 / bf16), N (videos per timing), REPS; STTM_LIB=dev (python -m sttm_amd.build --dev) for the col_abl ablation bits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib, get_quadtree_features, get_quadtree_features_batch
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
lib = _lib.load()
T, C = int(os.environ.get("T", "128")), int(os.environ.get("C", "1024"))
dt = torch.float32 if os.environ.get("DT", "f32") == "f32" else torch.bfloat16
N, REPS, P, CALL = int(os.environ.get("N", "1536")), int(os.environ.get("REPS", "3")), 8, 96
pool = [synth_video(T, C, 14, 14, seed=100 + i, dtype=dt, device=dev, gen_device=dev) for i in range(P)]
DEFAULT = dict(col_walk=1, col_frames=8, col_cap=0, col_pb=0, batch_streams=3, batch_sub=8)
if os.environ.get("STTM_LIB") == "dev":                 # the ablation bits (col_abl) exist in the development build only
    DEFAULT["col_abl"] = 0
_lib.configure(col_walk=0)
ref = [get_quadtree_features(v, 0.85, 0.55, 1) for v in pool]
torch.cuda.synchronize()


def run_once():
    vids = [pool[i % P] for i in range(CALL)]
    for _ in range(N // CALL):
        get_quadtree_features_batch(vids, 0.85, 0.55, 1)
    torch.cuda.synchronize()


def rate():
    t0 = time.perf_counter(); run_once()
    return N / (time.perf_counter() - t0)


settings = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [
    dict(col_walk=0), dict(), dict(col_frames=4), dict(col_frames=16), dict(col_frames=32),
    dict(batch_streams=4), dict(batch_streams=6), dict(batch_sub=4), dict(batch_sub=16), dict(col_walk=0, batch_streams=4)]
settings = [{k: int(v) for k, v in s.items()} for s in settings]
res = {i: [] for i in range(len(settings))}
for i, s in enumerate(settings):                     # correctness first
    _lib.configure(**{**DEFAULT, **s})
    out = get_quadtree_features_batch(pool + pool[:3], 0.85, 0.55, 1)
    torch.cuda.synchronize()
    for j, (f, n, t) in enumerate(out):
        rf, rn, rt = ref[j % P]
        assert s.get('col_abl') or (torch.equal(t, rt) and torch.equal(n, rn) and torch.equal(f, rf)), (s, j)
    run_once()
for rep in range(REPS):                              # interleaved rounds
    for i, s in enumerate(settings):
        _lib.configure(**{**DEFAULT, **s})
        res[i].append(rate())
_lib.configure(**DEFAULT)
print(f"T={T} C={C} {dt} N={N} build {_lib.build_tag()}")
for i, s in enumerate(settings):
    r = sorted(res[i])
    print(f"{str(s or 'default'):60s} median {r[len(r) // 2]:8.0f}  max {r[-1]:8.0f} videos/s")
