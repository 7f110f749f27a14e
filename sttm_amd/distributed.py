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
