"""Multi-GPU path on the device (`-m gpu`): ONE rank under torchrun with the nccl (RCCL) backend -- the launcher, the process
group, the sharding and both collectives of sttm_amd.distributed with the HIP merge doing the work -- and bench.py's own
self-spawn + --validate mode at a small size.  (World size 2 of the same code runs on CPU with gloo: test_sharding_gloo.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["STTM_REPO"])
from oracle import sttm_oracle as O
from sttm_amd import get_quadtree_features
from sttm_amd.distributed import gather_counts, gather_indices, shard_videos
from sttm_amd.synth import synth_video
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
assert dist.get_backend() == "nccl"
n_videos, T, H, W = 5, 6, 14, 14
ids = shard_videos(n_videos, world, rank)
counts, indices = [], []
for v in ids:
    f, n, t = get_quadtree_features(synth_video(T, 64, H, W, seed=v).to(dev), 0.85, 0.55, 1)
    counts.append(f.shape[0]); indices.append(t[:, 0] * (H * W) + t[:, 1] * W + t[:, 2])
full = gather_counts(ids, counts, n_videos, dev, dist)
idx = gather_indices(ids, indices, n_videos, T * H * W, dev, dist)
dist.barrier(device_ids=[int(os.environ["LOCAL_RANK"])])
for v in range(n_videos):
    ef, en, et = O.get_quadtree_features(synth_video(T, 64, H, W, seed=v), 0.85, 0.55, 1)
    assert int(full[v]) == ef.shape[0], (v, int(full[v]), ef.shape[0])
    exp = (et[:, 0] * (H * W) + et[:, 1] * W + et[:, 2]).to(torch.int32)
    row = idx[v].cpu()
    assert torch.equal(row[:exp.numel()], exp) and bool((row[exp.numel():] == -1).all()), v
dist.destroy_process_group()
print("DIST_OK")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script_args, env=None, timeout=600):
    e = dict(os.environ, STTM_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=REPO)


def test_one_rank_torchrun_through_both_collectives(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = _torchrun(1, [str(script)])
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_ranks_over_rccl_when_the_box_has_two_gpus(tmp_path):
    """The same worker with WORLD_SIZE 2, one rank per GPU, nccl = RCCL over xGMI: gather_counts / gather_indices with the HIP
    merge on both devices.  Runs wherever the box exposes >= 2 GPUs (the driver's multi-GPU node); skipped on 1-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (one rank per GPU)")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = _torchrun(2, [str(script)])
    assert r.returncode == 0 and r.stdout.count("DIST_OK") == 2, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_validate_mode_under_torchrun():
    """bench.py as the driver launches it (torchrun, one rank here), small sizes, with the index gather of --validate."""
    r = _torchrun(1, [os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--videos-per-step", "8", "--frames", "16",
                      "--profile-calls", "8", "--no-cpu-baseline", "--no-extensions", "--no-configs", "--validate"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["validate"]["cross_rank_index_match"] is True
    assert line["validate"]["index_rows_gathered"] == line["validate"]["videos"]
    assert line["roofline"]["bound"] == "hbm" and line["value"] > 0


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = _torchrun(1, [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--videos-per-step", "2", "--frames", "8"])
    assert r.returncode != 0 and "launcher started 1 ranks" in (r.stdout + r.stderr)


def test_bench_fails_fast_when_the_node_has_fewer_gpus_than_asked_for():
    """`python bench.py --gpus N` on a node with fewer than N GPUs: a clear message at once, before any launcher / rendezvous."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and f"this node has {n} GPU" in (r.stdout + r.stderr)
    # the value_definition / both-definitions fields of the JSON line: a tiny run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--videos-per-step", "16", "--frames", "16",
                        "--batch", "8", "--profile-calls", "8", "--no-cpu-baseline", "--no-configs"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "batch entry point" in line["value_definition"] and line["value_batch_entry_point"] == line["value"]
    assert line["value_one_call_per_video"] == line["dropin_one_call_per_video"]["value"] > 0
