"""Multi-GPU sharding of the merge: videos are independent units, so they are partitioned over the ranks
(one process per GPU) with NO data-path collective; the only exchange is the final all-gather of the
per-video merged-token counts (and, in validation mode, indices) over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  SURVEY.md section 8(e).
"""
import torch


def shard_videos(n_videos, world, rank, costs=None):
    """Video ids owned by `rank`.  Uniform costs -> round-robin; otherwise longest-processing-time-first
    greedy on `costs` (e.g. T*H*W per video), deterministic on every rank."""
    if costs is None:
        return list(range(rank, n_videos, world))
    order = sorted(range(n_videos), key=lambda i: (-costs[i], i))
    load = [0] * world
    owner = [0] * n_videos
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return [i for i in range(n_videos) if owner[i] == rank]


def gather_counts(local_ids, local_counts, n_videos, device, dist=None):
    """All ranks learn every video's merged-token count: returns an int32 tensor [n_videos].
    local_ids / local_counts: this rank's video ids and N' values.  One collective, tiny payload."""
    full = torch.zeros(n_videos, dtype=torch.int32, device=device)
    if len(local_ids):
        full[torch.as_tensor(local_ids, device=device, dtype=torch.long)] = torch.as_tensor(
            local_counts, device=device, dtype=torch.int32)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        # disjoint ownership: a SUM all-reduce of the one-hot-by-owner vector is an all-gather with static shape
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full


def gather_indices(local_ids, local_indices, n_videos, max_len, device, dist=None):
    """Validation mode of SURVEY 8(e)(ii): every rank learns every video's merged-token indices (t*H*W + y1*W + x1, int32),
    for the cross-rank parity check and token-ratio statistics.  Returns int32 [n_videos, max_len], rows padded with -1.
    local_indices: this rank's index tensors, in the order of local_ids.  ONE all-gather of a padded
    [videos per rank, 1 + max_len] block (column 0 = video id) behind a two-int MAX all-reduce that fixes the block height and
    carries every rank's argument check (an invalid call raises ValueError on ALL ranks instead of hanging the others);
    features never travel."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    # Local validation FIRST, and its verdict travels with the block height: a rank that raised before (or between) the
    # collectives would leave its peers hanging in them.  Cost of the call: two collectives (a two-int MAX all-reduce with one
    # host sync, then the all-gather).
    problem = None
    if len(local_ids) != len(local_indices):
        problem = f"{len(local_ids)} video ids but {len(local_indices)} index tensors"
    else:
        for vid, idx in zip(local_ids, local_indices):
            if idx.numel() > max_len:
                problem = f"video {vid}: {idx.numel()} indices do not fit max_len={max_len}"
                break
    # rows per rank = the LARGEST ownership of any rank: cost-balanced sharding (shard_videos(costs=...)) may give one rank
    # more than ceil(n_videos / world) videos, so the ranks agree on the block height with one MAX all-reduce
    per_rank = len(local_ids)
    failed = 1 if problem else 0
    if world > 1:
        most = torch.tensor([per_rank, failed], dtype=torch.int32, device=device)
        dist.all_reduce(most, op=dist.ReduceOp.MAX)
        per_rank, failed = (int(v) for v in most.tolist())
    if failed:            # on EVERY rank, after the collective
        raise ValueError(problem or "gather_indices: another rank reported invalid arguments (see its error)")
    block = torch.full((max(per_rank, 1), 1 + max_len), -1, dtype=torch.int32, device=device)
    for k, (vid, idx) in enumerate(zip(local_ids, local_indices)):
        block[k, 0] = vid
        block[k, 1:1 + idx.numel()] = idx.to(device=device, dtype=torch.int32)
    if world > 1:
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block)
        block = torch.cat(parts, dim=0)
    full = torch.full((n_videos, max_len), -1, dtype=torch.int32, device=device)
    owned = block[:, 0] >= 0
    full[block[owned, 0].long()] = block[owned, 1:]
    return full


class _NoSync:
    """Device hooks of run_sharded_job for a CPU world (the gloo tests): nothing to synchronise."""
    @staticmethod
    def synchronize():
        pass


def run_sharded_job(step_fn, steps, warmup, videos_per_step, rank, world, device, dist=None, sync=None, barrier_device_ids=None):
    """The rank logic of `bench.py` (one process per GPU; weak scaling: every rank runs `videos_per_step` videos per step).

    step_fn(step_index, sink): runs one step on this rank and appends the merged-token count N' of every video to `sink`.
    Order of events, identical on every rank:
      1. `warmup` untimed steps; with a process group also one warm `gather_counts` and one MAX all-reduce (RCCL builds its
         communicator and channels lazily, ~50 ms once) -- then barrier + device synchronise;
      2. t0; `steps` timed steps; the job's ONE exchange step: `gather_counts` of this rank's N' values (every rank must end up
         with all world * steps * videos_per_step of them, checked); device synchronise; barrier; t1;
      3. the MAX of (t1 - t0) over the ranks (one all-reduce) is the job's time.
    Returns dict(elapsed_s, videos, value = videos / elapsed_s, t_issue_s, counts = this rank's N' list, all_counts).
    `sync`: object with .synchronize() (torch.cuda on the GPU; nothing on CPU); `barrier_device_ids`: passed to dist.barrier on NCCL."""
    import time
    sync = sync or _NoSync
    multi = dist is not None and dist.is_initialized()
    K, V = int(steps), int(videos_per_step)

    def barrier():
        if multi:
            if barrier_device_ids is not None:
                dist.barrier(device_ids=barrier_device_ids)
            else:
                dist.barrier()

    sink = []
    for s in range(int(warmup)):
        step_fn(s, sink)
    if multi:
        ids_w = shard_videos(world * K * V, world, rank)
        cw = gather_counts(ids_w, [1] * len(ids_w), world * K * V, device, dist)
        assert int((cw > 0).sum()) == world * K * V
        tw = torch.zeros(1, dtype=torch.float64, device=device)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        barrier()
    sync.synchronize()
    sink = []
    barrier()
    sync.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        step_fn(s, sink)
    t_issue = time.perf_counter() - t0
    if len(sink) != K * V:
        raise RuntimeError(f"rank {rank}: the step function reported {len(sink)} videos, expected {K * V}")
    all_counts = None
    if multi:
        ids = shard_videos(world * K * V, world, rank)
        all_counts = gather_counts(ids, sink, world * K * V, device, dist)
        n_seen = int((all_counts > 0).sum())
        if n_seen != world * K * V:
            raise RuntimeError(f"rank {rank}: the count gather saw {n_seen} of {world * K * V} videos")
    sync.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    videos = world * K * V
    return {"elapsed_s": elapsed, "videos": videos, "value": videos / elapsed, "t_issue_s": t_issue, "counts": sink,
            "all_counts": all_counts}


# ---- host placement: a rank's launch thread spins on a pinned-memory word after every video, so it should sit on the CPUs next to
#      its GPU's PCIe root, and the ranks of one node must not share cores ---------------------------------------------------------

def parse_cpu_list(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def read_gpu_numa_topology(pci_addresses, sysfs_root="/sys"):
    """NUMA node of every GPU (by PCI address 'dddd:bb:dd.f') and the CPUs of those nodes, read from a sysfs tree.
    Returns (nodes: list of int or None per GPU, node_cpus: {node: [cpu, ...]}); a GPU whose node is not exposed (-1 / no file:
    containers often hide it) gets None."""
    import os
    nodes, node_cpus = [], {}
    for addr in pci_addresses:
        node = None
        try:
            with open(os.path.join(sysfs_root, "bus", "pci", "devices", addr, "numa_node")) as fh:
                n = int(fh.read().strip())
            if n >= 0:
                node = n
        except (OSError, ValueError):
            node = None
        nodes.append(node)
        if node is not None and node not in node_cpus:
            try:
                with open(os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")) as fh:
                    node_cpus[node] = parse_cpu_list(fh.read())
            except (OSError, ValueError):
                node_cpus[node] = []
    return nodes, node_cpus


def numa_cpu_sets(gpu_nodes, node_cpus, allowed=None):
    """CPU set of every local rank (rank i drives GPU i): the CPUs of its GPU's NUMA node that this process may use, and -- when
    several ranks' GPUs hang off the SAME node -- that node's CPUs dealt out in contiguous, DISJOINT shares (in rank order, sizes
    differing by at most one), so two ranks never spin on one core.  A rank whose node is unknown or has no usable CPU gets None
    (leave its affinity alone)."""
    sets = [None] * len(gpu_nodes)
    by_node = {}
    for r, n in enumerate(gpu_nodes):
        if n is not None:
            by_node.setdefault(n, []).append(r)
    for n, ranks in by_node.items():
        cpus = sorted(c for c in node_cpus.get(n, []) if allowed is None or c in allowed)
        if len(cpus) < len(ranks):
            continue                      # fewer usable CPUs than ranks on this node: do not pin (sharing is worse than floating)
        base, extra = divmod(len(cpus), len(ranks))
        at = 0
        for k, r in enumerate(ranks):
            size = base + (1 if k < extra else 0)
            sets[r] = cpus[at:at + size]
            at += size
    return sets
