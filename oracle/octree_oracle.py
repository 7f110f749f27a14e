"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's octree baseline
(`get_octree_features` / `octree_build`, token_merging_utils/octree_utils.py:293-389).

Parity pinned: checked against tests/golden/oct_*.npz, which tests/golden/make_golden_octree.py produced by running the
reference's own function on the CPU (float32; ATen has no CPU avg_pool3d for bfloat16, so 16-bit inputs are NOT pinned to
reference vectors -- the restatement rounds once per pyramid level, like the GPU operators the reference would run).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path never does.

The clip is cut into cubes of `side` frames (side = W, octree_utils.py:296-299); every cube gets a 3-D pyramid with the same
"first cell stays alone" rule per axis as the quadtree (:17-148), parents are compared with their 8 child slots -- slots that
do not exist alias child cell (0,0,0) of the SAME cube (:204-205 zeros + :262-266) -- and a parent whose 8 cosines all reach
the threshold is emitted whole (:281-289).  Emitted nodes are ordered by the leaf index of their first corner (:369-373).
Frames that do not fill a cube are merged per frame by the spatial quadtree (:375-378); clips shorter than one cube are
quadtree-only (:305-306).

Own structure: per axis the child table is closed form (`axis_split`), so a level is pooled with eight masked gathers, the
stop decision of EVERY cell is taken level by level (it does not depend on the frontier), and one top-down sweep of boolean
"alive" volumes picks the emitted cells.
"""
import math

import torch
import torch.nn.functional as F

from oracle.sttm_oracle import axis_split
from oracle import sttm_oracle as O


def side_levels(side):
    """[2, ..., side]: halve (ceil) until 2 (:312-316)."""
    sizes = [side]
    w = side
    guard = 0
    while w != 2:
        w = math.ceil(w / 2)
        sizes.insert(0, w)
        guard += 1
        if guard > 64:
            raise ValueError("side never reaches 2")
    return sizes


def _child_index(n_child):
    """LongTensors start[p], count[p] for one axis."""
    start, count = axis_split(n_child)
    return torch.tensor(start), torch.tensor(count)


def pool_cube_level(v):
    """v [B, s, s, s, C] -> [B, ceil(s/2)^3, C] grid: mean of the 1/2/4/8 children, float32 sum in (t, y, x) order,
    one rounding to the input dtype (:17-148)."""
    B, s, _, _, C = v.shape
    start, count = _child_index(s)
    n = start.numel()
    acc = torch.zeros((B, n, n, n, C), dtype=torch.float32)
    cnt = torch.zeros((n, n, n), dtype=torch.float32)
    vf = v.float()
    for dt in range(2):
        for dy in range(2):
            for dx in range(2):
                ok = (count[:, None, None] > dt) & (count[None, :, None] > dy) & (count[None, None, :] > dx)      # [n, n, n]
                it = (start + dt).clamp_max(s - 1)
                iy = (start + dy).clamp_max(s - 1)
                ix = (start + dx).clamp_max(s - 1)
                g = vf[:, it][:, :, iy][:, :, :, ix]                                                          # [B, n, n, n, C]
                acc = torch.where(ok[None, :, :, :, None], acc + g, acc)
                cnt = cnt + ok.float()
    return (acc / cnt[None, :, :, :, None]).to(v.dtype)


def stop_volume(parent, child, threshold):
    """parent [B, n, n, n, C], child [B, s, s, s, C] -> bool [B, n, n, n]: all 8 slot cosines >= threshold; slots without a
    child use child cell (0, 0, 0) of the same cube (:262-281)."""
    B, s = child.shape[0], child.shape[1]
    start, count = _child_index(s)
    pf = parent.float()
    cf = child.float()
    stop = torch.ones(parent.shape[:4], dtype=torch.bool)
    for dt in range(2):
        for dy in range(2):
            for dx in range(2):
                ok = (count[:, None, None] > dt) & (count[None, :, None] > dy) & (count[None, None, :] > dx)
                it = torch.where(count > dt, start + dt, torch.zeros_like(start))
                iy = torch.where(count > dy, start + dy, torch.zeros_like(start))
                ix = torch.where(count > dx, start + dx, torch.zeros_like(start))
                g = cf[:, it][:, :, iy][:, :, :, ix]
                alias = cf[:, 0, 0, 0][:, None, None, None, :].expand_as(g)
                g = torch.where(ok[None, :, :, :, None], g, alias)
                sim = F.cosine_similarity(pf, g, dim=-1)
                stop &= sim >= threshold
    return stop


def octree_cubes(x_bsssc, threshold, root_level):
    """x [B, side, side, side, C] -> features [N, C] ordered by the first-corner leaf index."""
    B, side = x_bsssc.shape[0], x_bsssc.shape[1]
    sizes = side_levels(side)
    target = sizes[root_level]                          # IndexError like the reference for an out-of-range root level
    levels = [x_bsssc]
    while levels[0].shape[1] != target:
        levels.insert(0, pool_cube_level(levels[0]))
    L = len(levels)
    # first-corner leaf coordinate of every cell, per level and axis
    first = [None] * L
    first[L - 1] = torch.arange(side)
    for l in range(L - 2, -1, -1):
        start, _ = _child_index(levels[l + 1].shape[1])
        first[l] = first[l + 1][start]
    alive = torch.ones(levels[0].shape[:4], dtype=torch.bool)
    feats, keys = [], []
    for l in range(L):
        n = levels[l].shape[1]
        stop = stop_volume(levels[l], levels[l + 1], threshold) if l < L - 1 else torch.ones_like(alive)
        emit = alive & stop
        b, t, y, xx = emit.nonzero(as_tuple=True)
        feats.append(levels[l][b, t, y, xx])
        keys.append(((b * side + first[l][t]) * side + first[l][y]) * side + first[l][xx])
        if l == L - 1:
            break
        split = alive & ~stop
        s = levels[l + 1].shape[1]
        start, count = _child_index(s)
        nxt = torch.zeros(levels[l + 1].shape[:4], dtype=torch.bool)
        for dt in range(2):
            for dy in range(2):
                for dx in range(2):
                    ok = (count[:, None, None] > dt) & (count[None, :, None] > dy) & (count[None, None, :] > dx)
                    sel = split & ok[None]
                    b, t, y, xx = sel.nonzero(as_tuple=True)
                    nxt[b, start[t] + dt, start[y] + dy, start[xx] + dx] = True
        alive = nxt
    feats = torch.cat(feats, 0)
    keys = torch.cat(keys, 0)
    return feats[torch.argsort(keys)]


def get_octree_features(_video_feature, threshold, root_level=0):
    """_video_feature: logical [T, C, H, W].  Returns features [N, C] in the input dtype."""
    T, C, H, W = _video_feature.shape
    side = W
    n_cube = T // side
    if n_cube == 0:
        return O.get_quadtree_features(_video_feature, threshold, -1.0, root_level)[0]
    if H != W:
        raise RuntimeError("the octree needs square frames (cube side = W)")
    drop = T % side
    body = _video_feature[:T - drop] if drop else _video_feature
    cubes = body.permute(0, 2, 3, 1).reshape(n_cube, side, H, W, C)
    out = octree_cubes(cubes, threshold, root_level)
    if drop:
        rem = O.get_quadtree_features(_video_feature[T - drop:], threshold, -1.0, root_level)[0]
        out = torch.cat([out, rem], 0)
    return out
