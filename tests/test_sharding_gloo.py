"""N > 1 path on CPU: two gloo ranks shard the videos, each merges its own (oracle as the compute stand-in),
and the final gather reproduces the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sttm_amd.distributed import gather_counts, gather_indices, shard_videos


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frames(v, costs):
    return 3 if costs is None else costs[v]


def _worker(rank, world, port, n_videos, out_dir, costs=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import sttm_oracle as O
    from sttm_amd.synth import synth_video
    ids = shard_videos(n_videos, world, rank, costs)
    counts, indices = [], []
    for v in ids:
        x = synth_video(_frames(v, costs), 16, 14, 14, seed=v)
        f, _, t = O.get_quadtree_features(x, 0.85, 0.55, 1)
        counts.append(f.shape[0])
        indices.append(t[:, 0] * 196 + t[:, 1] * 14 + t[:, 2])
    full = gather_counts(ids, counts, n_videos, torch.device("cpu"), dist)
    idx = gather_indices(ids, indices, n_videos, (3 if costs is None else max(costs)) * 196, torch.device("cpu"), dist)
    dist.barrier()
    torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    torch.save(idx, os.path.join(out_dir, f"i{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    n_videos, world = 5, 2
    mp.spawn(_worker, args=(world, _free_port(), n_videos, str(tmp_path)), nprocs=world, join=True)
    from oracle import sttm_oracle as O
    from sttm_amd.synth import synth_video
    outs = [O.get_quadtree_features(synth_video(3, 16, 14, 14, seed=v), 0.85, 0.55, 1) for v in range(n_videos)]
    expect = torch.tensor([o[0].shape[0] for o in outs], dtype=torch.int32)
    expect_idx = torch.full((n_videos, 3 * 196), -1, dtype=torch.int32)
    for v, (_, _, t) in enumerate(outs):
        expect_idx[v, :t.shape[0]] = t[:, 0] * 196 + t[:, 1] * 14 + t[:, 2]
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert torch.equal(got, expect)
        # the padded index gather of the validation mode: every rank holds every video's merged-token indices
        assert torch.equal(torch.load(os.path.join(str(tmp_path), f"i{r}.pt")), expect_idx)


def test_cost_balanced_sharding_through_the_index_gather(tmp_path):
    """Mixed clip lengths: LPT sharding gives rank 1 three of the four videos (more than ceil(4 / 2)); the padded index gather
    must size its block from the largest ownership (round-2 advisor finding: it raised IndexError)."""
    costs = [9, 1, 1, 1]                       # frames per video = its cost
    n_videos, world = len(costs), 2
    assert [len(shard_videos(n_videos, world, r, costs)) for r in range(world)] == [1, 3]
    mp.spawn(_worker, args=(world, _free_port(), n_videos, str(tmp_path), costs), nprocs=world, join=True)
    from oracle import sttm_oracle as O
    from sttm_amd.synth import synth_video
    outs = [O.get_quadtree_features(synth_video(costs[v], 16, 14, 14, seed=v), 0.85, 0.55, 1) for v in range(n_videos)]
    expect = torch.tensor([o[0].shape[0] for o in outs], dtype=torch.int32)
    expect_idx = torch.full((n_videos, max(costs) * 196), -1, dtype=torch.int32)
    for v, (_, _, t) in enumerate(outs):
        expect_idx[v, :t.shape[0]] = t[:, 0] * 196 + t[:, 1] * 14 + t[:, 2]
    for r in range(world):
        assert torch.equal(torch.load(os.path.join(str(tmp_path), f"r{r}.pt")), expect)
        assert torch.equal(torch.load(os.path.join(str(tmp_path), f"i{r}.pt")), expect_idx)


def _bad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 1 passes an index tensor longer than max_len; rank 0's arguments are fine
    idx = [torch.arange(20 if rank == 1 else 4, dtype=torch.int32)]
    try:
        gather_indices([rank], idx, 2, 8, torch.device("cpu"), dist)
        verdict = "returned"
    except ValueError as e:
        verdict = "ValueError: " + str(e)
    with open(os.path.join(out_dir, f"v{rank}.txt"), "w") as fh:
        fh.write(verdict)
    dist.barrier()                 # both ranks are still in step: nobody is stuck inside a collective
    dist.destroy_process_group()


def test_invalid_arguments_on_one_rank_raise_on_every_rank(tmp_path):
    """Round-3 advisor finding: a rank that raised before / between the two collectives of gather_indices left its peers hanging.
    The argument check now travels with the MAX all-reduce and every rank raises after it."""
    mp.spawn(_bad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    v0 = open(os.path.join(str(tmp_path), "v0.txt")).read()
    v1 = open(os.path.join(str(tmp_path), "v1.txt")).read()
    assert v0.startswith("ValueError") and "another rank" in v0
    assert v1.startswith("ValueError") and "do not fit max_len" in v1


def test_gather_indices_rejects_mismatched_lists():
    with pytest.raises(ValueError):
        gather_indices([0, 1], [torch.zeros(3, dtype=torch.int32)], 2, 8, torch.device("cpu"))


def test_shard_videos_partitions():
    for world in (1, 2, 4, 8):
        owned = [shard_videos(19, world, r) for r in range(world)]
        flat = sorted(i for o in owned for i in o)
        assert flat == list(range(19))
    costs = [128, 64, 180, 32, 128, 64, 16, 180]
    owned = [shard_videos(len(costs), 3, r, costs) for r in range(3)]
    assert sorted(i for o in owned for i in o) == list(range(len(costs)))
    loads = [sum(costs[i] for i in o) for o in owned]
    assert max(loads) - min(loads) <= max(costs)
