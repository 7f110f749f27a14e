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


def _job_worker(rank, world, port, out_dir):
    """bench.py's rank logic (sttm_amd.distributed.run_sharded_job) with the CPU oracle as the compute stand-in."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import sttm_oracle as O
    from sttm_amd.distributed import run_sharded_job
    from sttm_amd.synth import synth_video
    K, V = 2, 3
    calls = []

    def step(s, sink):
        calls.append(s)
        for v in range(V):
            # weak scaling: every rank has its own videos; seed = global video id of the round-robin sharding
            gid = (s * V + v) * world + rank
            f, _, _ = O.get_quadtree_features(synth_video(2, 16, 14, 14, seed=gid), 0.85, 0.55, 1)
            sink.append(f.shape[0])
    job = run_sharded_job(step, K, 1, V, rank, world, torch.device("cpu"), dist)
    torch.save({"job": {k: job[k] for k in ("elapsed_s", "videos", "value", "counts")}, "all": job["all_counts"], "calls": calls},
               os.path.join(out_dir, f"j{rank}.pt"))
    dist.destroy_process_group()


def test_bench_rank_logic_with_two_gloo_ranks(tmp_path):
    """The timed region of bench.py -- warm-up steps, warm collectives, exactly K timed steps between barriers, the count gather as the
    job's one exchange step, MAX of the elapsed time over the ranks -- run by two CPU ranks."""
    world = 2
    mp.spawn(_job_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import sttm_oracle as O
    from sttm_amd.synth import synth_video
    res = [torch.load(os.path.join(str(tmp_path), f"j{r}.pt")) for r in range(world)]
    K, V = 2, 3
    expect = torch.tensor([O.get_quadtree_features(synth_video(2, 16, 14, 14, seed=g), 0.85, 0.55, 1)[0].shape[0] for g in range(world * K * V)],
                          dtype=torch.int32)
    for r in range(world):
        assert res[r]["calls"] == [0, 0, 1]                                 # one warm-up step, then exactly K timed steps
        assert res[r]["job"]["videos"] == world * K * V
        assert torch.equal(res[r]["all"], expect)                            # every rank holds every video's count after the gather
        assert res[r]["job"]["counts"] == expect[r::world].tolist()
    # MAX over the ranks: both ranks report the SAME elapsed time and the same whole-job value
    assert res[0]["job"]["elapsed_s"] == res[1]["job"]["elapsed_s"] and res[0]["job"]["value"] == res[1]["job"]["value"]


def test_eight_ranks_mixed_clip_lengths_lpt_sharding_and_the_padded_index_gather(tmp_path):
    """The node the north star names: 8 ranks.  19 videos of mixed lengths, LPT sharding on the frame counts (uneven ownership: some ranks
    own three videos, some two), the count gather and the padded index gather -- every rank ends up with every video's result."""
    costs = [9, 1, 4, 2, 6, 1, 1, 3, 5, 2, 2, 1, 7, 1, 3, 2, 1, 4, 1]
    n_videos, world = len(costs), 8
    owned = [shard_videos(n_videos, world, r, costs) for r in range(world)]
    assert sorted(i for o in owned for i in o) == list(range(n_videos)) and len({len(o) for o in owned}) > 1
    loads = [sum(costs[i] for i in o) for o in owned]
    assert max(loads) - min(loads) <= max(costs)
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


def test_bench_rank_logic_with_eight_gloo_ranks(tmp_path):
    """bench.py --gpus 8 without the GPUs: run_sharded_job on 8 CPU ranks (weak scaling: 8 x K x V videos, MAX of the elapsed time over the
    ranks, one count gather)."""
    world = 8
    mp.spawn(_job_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import sttm_oracle as O
    from sttm_amd.synth import synth_video
    res = [torch.load(os.path.join(str(tmp_path), f"j{r}.pt")) for r in range(world)]
    K, V = 2, 3
    expect = torch.tensor([O.get_quadtree_features(synth_video(2, 16, 14, 14, seed=g), 0.85, 0.55, 1)[0].shape[0] for g in range(world * K * V)],
                          dtype=torch.int32)
    for r in range(world):
        assert res[r]["calls"] == [0, 0, 1] and res[r]["job"]["videos"] == world * K * V
        assert torch.equal(res[r]["all"], expect) and res[r]["job"]["counts"] == expect[r::world].tolist()
    assert len({res[r]["job"]["elapsed_s"] for r in range(world)}) == 1 and len({res[r]["job"]["value"] for r in range(world)}) == 1


def test_run_sharded_job_single_process_needs_no_process_group():
    from sttm_amd.distributed import run_sharded_job
    seen = []
    job = run_sharded_job(lambda s, sink: (seen.append(s), sink.extend([7, 7]))[0], 3, 2, 2, 0, 1, torch.device("cpu"))
    assert seen == [0, 1, 0, 1, 2] and job["videos"] == 6 and job["counts"] == [7] * 6 and job["all_counts"] is None
    with pytest.raises(RuntimeError, match="expected"):
        run_sharded_job(lambda s, sink: sink.append(1), 1, 0, 2, 0, 1, torch.device("cpu"))


def _fake_sysfs(root, gpus, nodes):
    """gpus: {pci address: numa node}, nodes: {node: cpulist text}"""
    for addr, node in gpus.items():
        d = os.path.join(root, "bus", "pci", "devices", addr)
        os.makedirs(d)
        with open(os.path.join(d, "numa_node"), "w") as fh:
            fh.write(f"{node}\n")
    for node, text in nodes.items():
        d = os.path.join(root, "devices", "system", "node", f"node{node}")
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as fh:
            fh.write(text + "\n")


def test_numa_pinning_gives_ranks_on_one_node_disjoint_cpu_sets(tmp_path):
    """8 GPUs on 2 NUMA nodes (the MI355X host layout): the four ranks of a node split its CPUs into disjoint contiguous shares."""
    from sttm_amd.distributed import numa_cpu_sets, parse_cpu_list, read_gpu_numa_topology
    addrs = [f"0000:{b:02x}:00.0" for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]
    _fake_sysfs(str(tmp_path), {a: (0 if i < 4 else 1) for i, a in enumerate(addrs)}, {0: "0-63,128-191", 1: "64-127,192-255"})
    nodes, node_cpus = read_gpu_numa_topology(addrs, str(tmp_path))
    assert nodes == [0, 0, 0, 0, 1, 1, 1, 1] and len(node_cpus[0]) == 128 and node_cpus[1][0] == 64
    sets = numa_cpu_sets(nodes, node_cpus)
    for a in range(8):
        assert sets[a] and set(sets[a]) <= set(node_cpus[nodes[a]])
        for b in range(a + 1, 8):
            assert not set(sets[a]) & set(sets[b]), (a, b)                   # no two ranks share a core
    assert [len(x) for x in sets] == [32] * 8
    assert sorted(c for x in sets[:4] for c in x) == node_cpus[0]            # the node is used completely
    # a restricted affinity mask (cgroup): only the allowed CPUs are dealt out, still disjoint
    sets = numa_cpu_sets(nodes, node_cpus, allowed=set(range(0, 10)) | set(range(64, 70)))
    assert [len(x) for x in sets[:4]] == [3, 3, 2, 2] and [len(x) for x in sets[4:]] == [2, 2, 1, 1]
    assert not set(sets[0]) & set(sets[1])
    # fewer usable CPUs than ranks on a node: nobody on that node is pinned (sharing a core is worse than floating)
    sets = numa_cpu_sets(nodes, node_cpus, allowed={0, 1, 64, 65, 66, 67})
    assert sets[:4] == [None] * 4 and [len(x) for x in sets[4:]] == [1, 1, 1, 1]
    # hidden topology (containers report -1 or have no file): no pinning, no exception
    _fake_sysfs(str(tmp_path / "c"), {"0000:05:00.0": -1}, {})
    nodes, node_cpus = read_gpu_numa_topology(["0000:05:00.0", "0000:06:00.0"], str(tmp_path / "c"))
    assert nodes == [None, None] and numa_cpu_sets(nodes, node_cpus) == [None, None]
    assert parse_cpu_list("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and parse_cpu_list("") == []
