"""CPU oracle for the STTM token-merging hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain PyTorch on CPU tensors, vectorised) of the algorithm
the reference implements in `token_merging_utils/` (HYUNJS/STTM).  It exists to *check* the HIP path:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product
package (`sttm_amd/`) never imports, calls or falls back to anything in `oracle/`.

Parity pinning: the reference ships no tests or golden vectors for this path, so
this oracle is pinned against outputs of the reference itself: `tests/golden/make_golden.py` imports the
reference from /root/reference in the build container, runs it on seeded inputs, and commits
inputs + outputs as `tests/golden/*.npz`; `tests/test_oracle_golden.py` requires this file to reproduce
every one of them (indices bit-exact, CPU fp32/bf16 features bit-exact), plus the hand-checkable
known-answer cases of SURVEY.md Appendix B.

Every function cites the reference lines (relative to /root/reference) whose behaviour it restates.
Layout convention here is channels-last ([T, h, w, C]) with *flat* per-frame cell ids; the reference
works on [T, C, h, w] views and (t, y, x) coordinate triplets.
"""
import math

import torch
import torch.nn.functional as F

EPS_SPATIAL = 1e-8   # F.cosine_similarity default eps (quadtree_builder.py:61)
EPS_TEMPORAL = 1e-8  # added to the norm, not clamped (quadtree_temporal_merger.py:62)


# ----------------------------------------------------------------------------------------------
# geometry: level sizes, split tables, boxes            (integer-only, data independent)
# ----------------------------------------------------------------------------------------------

def level_sizes(H, W):
    """Coarse-to-fine list of (h, w); halves (ceil) until EITHER side equals 2.
    quadtree_builder.py:101-106."""
    if H < 1 or W < 1 or (H == 1 and W == 1):
        raise ValueError("degenerate token grid")
    sizes = [(H, W)]
    h, w = H, W
    guard = 0
    while h != 2 and w != 2:
        w = math.ceil(w / 2)
        h = math.ceil(h / 2)
        sizes.insert(0, (h, w))
        guard += 1
        if guard > 64:
            raise ValueError("token grid never reaches a side of 2 (the reference loops forever here)")
    return sizes


def axis_split(n):
    """Children of each parent along one axis when a side of n cells is halved.
    even n: [0,1],[2,3],...   odd n: [0],[1,2],[3,4],...  (first cell stays alone)
    quadtree_spatial_merger.py:38-54 (features) and :155-271 (index tables).
    Returns (start, count) int lists of length ceil(n/2)."""
    if n % 2 == 0:
        start = [2 * i for i in range(n // 2)]
        count = [2] * (n // 2)
    else:
        start = [0] + [2 * i - 1 for i in range(1, (n + 1) // 2)]
        count = [1] + [2] * ((n - 1) // 2)
    return start, count


class Geometry:
    """All data-independent tables of one (H, W, root_level) configuration."""

    def __init__(self, H, W, root_level):
        sizes = level_sizes(H, W)
        target_w = sizes[root_level][1]          # IndexError for out-of-range levels, like the reference
        # pool until the WIDTH matches (quadtree_builder.py:111) -- width only, quirk Q5
        dims = [(H, W)]
        while dims[0][1] != target_w:
            h, w = dims[0]
            dims.insert(0, (math.ceil(h / 2), math.ceil(w / 2)))
            if len(dims) > 64:
                raise ValueError("pyramid never reaches the requested root level")
        self.H, self.W = H, W
        self.dims = dims                          # coarse -> fine; dims[-1] == (H, W)
        self.n_level = len(dims)
        # leaf-unit boxes per level, separable in y and x
        ys = [(list(range(H)), [y + 1 for y in range(H)])]
        xs = [(list(range(W)), [x + 1 for x in range(W)])]
        self.row_split, self.col_split = [], []   # index l: split of level l+1 rows into level l rows
        for lvl in range(self.n_level - 1, 0, -1):
            h, w = dims[lvl]
            rs, rc = axis_split(h)
            cs, cc = axis_split(w)
            self.row_split.insert(0, (rs, rc))
            self.col_split.insert(0, (cs, cc))
            y1, y2 = ys[0]
            x1, x2 = xs[0]
            ys.insert(0, ([y1[s] for s in rs], [y2[s + c - 1] for s, c in zip(rs, rc)]))
            xs.insert(0, ([x1[s] for s in cs], [x2[s + c - 1] for s, c in zip(cs, cc)]))
        self.ys, self.xs = ys, xs

    def child_table(self, lvl):
        """Flat child ids [h*w, 4] (slot k = 2*dy + dx) into level lvl+1 and their validity.
        Invalid slots point at child cell 0 == (0, 0): quirk Q1
        (quadtree_spatial_merger.py:190-191 zero-initialises the coordinates)."""
        h, w = self.dims[lvl]
        _, wc = self.dims[lvl + 1]
        rs, rc = self.row_split[lvl]
        cs, cc = self.col_split[lvl]
        ids = torch.zeros(h * w, 4, dtype=torch.int64)
        ok = torch.zeros(h * w, 4, dtype=torch.bool)
        for i in range(h):
            for j in range(w):
                for dy in range(rc[i]):
                    for dx in range(cc[j]):
                        k = 2 * dy + dx
                        ids[i * w + j, k] = (rs[i] + dy) * wc + (cs[j] + dx)
                        ok[i * w + j, k] = True
        return ids, ok

    def boxes(self, lvl):
        """[h*w, 4] int32 (y1, x1, y2, x2) in leaf units for every cell of level lvl."""
        h, w = self.dims[lvl]
        y1, y2 = self.ys[lvl]
        x1, x2 = self.xs[lvl]
        b = torch.empty(h, w, 4, dtype=torch.int32)
        b[:, :, 0] = torch.tensor(y1, dtype=torch.int32)[:, None]
        b[:, :, 1] = torch.tensor(x1, dtype=torch.int32)[None, :]
        b[:, :, 2] = torch.tensor(y2, dtype=torch.int32)[:, None]
        b[:, :, 3] = torch.tensor(x2, dtype=torch.int32)[None, :]
        return b.reshape(h * w, 4)


# ----------------------------------------------------------------------------------------------
# pyramid                                               (fp; one rounding per level in input dtype)
# ----------------------------------------------------------------------------------------------

def pool_level(x, mode):
    """[T, h, w, C] -> [T, ceil(h/2), ceil(w/2), C].
    avg: uniform mean over the 1, 2 or 4 children (quadtree_spatial_merger.py:9-56);
    sum: plain sum (F.lp_pool2d p=1, :58-86), which only exists for equal parities (quirk Q6).
    4-child blocks are accumulated row-major ((a+b)+c)+d in fp32 like avg_pool2d does."""
    T, h, w, C = x.shape
    nh, nw = math.ceil(h / 2), math.ceil(w / 2)
    oy, ox = h % 2, w % 2
    if mode == "sum" and oy != ox:
        raise RuntimeError("sum pooling (weighted_avg=True) is undefined for mixed-parity grids "
                           f"({h}x{w}); the reference raises here too")
    lowp = x.dtype in (torch.bfloat16, torch.float16)
    xf = x.float() if lowp else x
    out = torch.empty(T, nh, nw, C, dtype=xf.dtype)
    body = xf[:, oy:, ox:]
    s = body[:, 0::2, 0::2] + body[:, 0::2, 1::2] + body[:, 1::2, 0::2] + body[:, 1::2, 1::2]
    out[:, oy:, ox:] = s / 4 if mode == "avg" else s
    if oy:
        row = xf[:, 0, ox:]
        s = row[:, 0::2] + row[:, 1::2]
        out[:, 0, ox:] = s / 2 if mode == "avg" else s
    if ox:
        col = xf[:, oy:, 0]
        s = col[:, 0::2] + col[:, 1::2]
        out[:, oy:, 0] = s / 2 if mode == "avg" else s
    if oy and ox:
        out[:, 0, 0] = xf[:, 0, 0]
    return out.to(x.dtype) if lowp else out


def build_pyramid(x_thwc, geom, mode):
    """Coarse -> fine list of [T, h*w, C] maps (quadtree_builder.py:109-125)."""
    maps = [x_thwc]
    for _ in range(geom.n_level - 1):
        maps.insert(0, pool_level(maps[0], mode))
    return [m.reshape(m.shape[0], -1, m.shape[-1]) for m in maps]


# ----------------------------------------------------------------------------------------------
# spatial stage: top-down split                         (quadtree_builder.py:18-83, :177-209)
# ----------------------------------------------------------------------------------------------

def _cosine(p, c, head_dim):
    """p [n,1,C], c [n,4,C] fp32 -> [n,4]; per-head mean when head_dim is given
    (quadtree_builder.py:58-66)."""
    if head_dim is None:
        return F.cosine_similarity(p, c, dim=-1, eps=EPS_SPATIAL)
    n = p.shape[0]
    ph = p.reshape(n, 1, -1, head_dim)
    ch = c.reshape(n, 4, -1, head_dim)
    return F.cosine_similarity(ph, ch, dim=-1, eps=EPS_SPATIAL).mean(dim=-1)


def spatial_split(maps, geom, threshold, head_dim=None, extra_maps=()):
    """Returns (features [N, C] input dtype, tlbr [N, 5] int32) sorted by (t, y1, x1); with `extra_maps` (pyramids
    of tensors that ride along, e.g. RoPE cos/sin, quadtree_builder.py:75-81) also their emitted rows."""
    T = maps[0].shape[0]
    h0, w0 = geom.dims[0]
    t_idx = torch.arange(T).repeat_interleave(h0 * w0)
    cell = torch.arange(h0 * w0).repeat(T)
    feats, boxes = [], []
    extras = [[] for _ in extra_maps]
    for lvl in range(geom.n_level):
        fmap = maps[lvl]
        box_tab = geom.boxes(lvl)
        parent = fmap[t_idx, cell]                                   # [n, C]
        pbox = torch.cat([t_idx[:, None].to(torch.int32), box_tab[cell]], dim=1)
        if lvl == geom.n_level - 1:                                  # leaves are always emitted (:26-37)
            feats.append(parent)
            boxes.append(pbox)
            for e, em in zip(extras, extra_maps):
                e.append(em[lvl][t_idx, cell])
            break
        ids, ok = geom.child_table(lvl)
        kid = ids[cell]                                              # [n, 4]
        kid_ok = ok[cell]
        child = maps[lvl + 1][t_idx[:, None], kid]                   # [n, 4, C]  (invalid -> cell (t,0,0))
        sim = _cosine(parent[:, None].float(), child.float(), head_dim)
        stop = (sim >= threshold).all(dim=-1)                        # over all 4 slots incl. invalid (:68)
        feats.append(parent[stop])
        boxes.append(pbox[stop])
        for e, em in zip(extras, extra_maps):
            e.append(em[lvl][t_idx, cell][stop])
        go = (~stop)[:, None] & kid_ok
        t_idx = t_idx[:, None].expand(-1, 4)[go]
        cell = kid[go]
    feats = torch.cat(feats)
    boxes = torch.cat(boxes)
    key = (boxes[:, 0].long() * geom.H + boxes[:, 1]) * geom.W + boxes[:, 2]
    order = torch.argsort(key)                                       # keys are unique
    if extra_maps:
        return feats[order], boxes[order], [torch.cat(e)[order] for e in extras]
    return feats[order], boxes[order]


# ----------------------------------------------------------------------------------------------
# temporal stage
# ----------------------------------------------------------------------------------------------

def frame_offsets(tlbr):
    """Start index of each run of equal t in the sorted node list, plus N
    (quadtree_temporal_merger.py:13-16)."""
    t = tlbr[:, 0]
    starts = torch.nonzero(t[1:] != t[:-1]).flatten() + 1
    return torch.cat([torch.zeros(1, dtype=torch.int64), starts, torch.tensor([t.numel()])])


def candidate_pairs(tlbr):
    """All (i in frame k, j in frame k+1) whose boxes nest either way, inclusive compare
    (quadtree_temporal_merger.py:8-56).  [L, 2] int64, column 0 = earlier frame (dst)."""
    off = frame_offsets(tlbr)
    n_frame = off.numel() - 1
    if n_frame < 2:
        return torch.zeros(0, 2, dtype=torch.int64)
    cnt = off[1:] - off[:-1]
    M = int(cnt.max())
    slot = torch.arange(M)[None, :]
    live = slot < cnt[:, None]                                       # [F, M]
    gidx = (off[:-1, None] + slot).clamp(max=tlbr.shape[0] - 1)
    box = tlbr[:, 1:][gidx]                                          # [F, M, 4]
    a, b = box[:-1, :, None, :], box[1:, None, :, :]
    a_has_b = (a[..., 0] <= b[..., 0]) & (a[..., 1] <= b[..., 1]) & (a[..., 2] >= b[..., 2]) & (a[..., 3] >= b[..., 3])
    b_has_a = (a[..., 0] >= b[..., 0]) & (a[..., 1] >= b[..., 1]) & (a[..., 2] <= b[..., 2]) & (a[..., 3] <= b[..., 3])
    hit = (a_has_b | b_has_a) & live[:-1, :, None] & live[1:, None, :]
    f, i, j = torch.nonzero(hit, as_tuple=True)
    return torch.stack([gidx[f, i], gidx[f + 1, j]], dim=1)


def candidate_pairs_by_owner(tlbr, H, W):
    """Independent cross-check of `candidate_pairs` (NOT the reference's method): because the nodes
    of a frame tile it with nested-or-disjoint boxes, two nodes of consecutive frames nest iff they
    share a leaf.  Valid only while every frame is fully tiled (i.e. straight after the spatial stage)."""
    N = tlbr.shape[0]
    T = int(tlbr[:, 0].max()) + 1
    owner = torch.full((T, H, W), -1, dtype=torch.int64)
    for n in range(N):
        t, y1, x1, y2, x2 = tlbr[n].tolist()
        owner[t, y1:y2, x1:x2] = n
    pr = torch.stack([owner[:-1].reshape(-1), owner[1:].reshape(-1)], dim=1)
    return torch.unique(pr, dim=0)


def unit_rows(x, head_dim=None):
    """x.float() / (||x|| + 1e-8), optionally per head (quadtree_temporal_merger.py:60-66)."""
    xf = x.float()
    if head_dim is not None:
        xf = xf.reshape(xf.shape[0], -1, head_dim)
    return xf / (xf.norm(dim=-1, keepdim=True) + EPS_TEMPORAL)


def pair_similarity(xn, pairs):
    s = (xn[pairs[:, 0]] * xn[pairs[:, 1]]).sum(dim=-1)
    return s.mean(dim=-1) if s.dim() == 2 else s


def keep_similar_pairs(x, pairs, temporal_thresh, head_dim=None):
    """quadtree_temporal_merger.py:58-73."""
    sim = pair_similarity(unit_rows(x, head_dim), pairs)
    return pairs[sim >= temporal_thresh]


def slow_pairs(x, tlbr, temporal_thresh):
    """slow_ver (quadtree_temporal_merger.py:75-121): per frame pair keep sim >= thr, order by sim
    descending, and drop an edge only when its src equals the src of the edge just before it in that
    order (adjacent-duplicate removal, not an arg-max per src).  No per-head variant."""
    off = frame_offsets(tlbr).tolist()
    xn = unit_rows(x)
    out = []
    for k in range(len(off) - 2):
        a = tlbr[off[k]:off[k + 1], 1:]
        b = tlbr[off[k + 1]:off[k + 2], 1:]
        d = a[:, None, :] - b[None, :, :]
        a_has_b = (d[..., :2] <= 0).all(-1) & (d[..., 2:] >= 0).all(-1)
        b_has_a = (d[..., :2] >= 0).all(-1) & (d[..., 2:] <= 0).all(-1)
        pr = torch.nonzero(a_has_b | b_has_a).to(torch.int32)
        pr[:, 0] += off[k]
        pr[:, 1] += off[k + 1]
        sim = (xn[pr[:, 0]] * xn[pr[:, 1]]).sum(dim=-1)
        sel = torch.nonzero(sim >= temporal_thresh).flatten()
        sim, pr = sim[sel], pr[sel]
        _, order = sim.sort(descending=True)
        dst, src = pr[order, 0], pr[order, 1]
        keep = torch.ones_like(src, dtype=torch.bool)
        keep[1:] = src[1:] != src[:-1]
        out.append(torch.stack([dst[keep], src[keep]], dim=1))
    if not out:
        return torch.zeros(0, 2, dtype=torch.int64)
    return torch.cat(out).to(torch.int64)


def propagate_labels(pairs, N):
    """Synchronous hook + pointer-jump iteration that stops on idempotency -- NOT full connected
    components (quirk Q2).  quadtree_temporal_merger.py:223-269."""
    rep = torch.arange(N, dtype=torch.int32)
    d, s = pairs[:, 0].long(), pairs[:, 1].long()
    iters = 0
    while True:
        m = torch.minimum(rep[d], rep[s])
        rep.scatter_reduce_(0, d, m, reduce="amin")
        rep.scatter_reduce_(0, s, m, reduce="amin")
        rep = rep[rep.long()]
        iters += 1
        if torch.equal(rep, rep[rep.long()]):
            return rep, iters


def aggregate_extra(v, rep, npatch, weighted):
    """Position embeddings ride along the groups (quadtree_temporal_merger.py:153-169)."""
    N = v.shape[0]
    rep = rep.to(torch.int32)
    acc = torch.zeros_like(v).index_add_(0, rep, v)
    cnt = torch.zeros(N, dtype=torch.int32).index_add_(0, rep, torch.ones(N, dtype=torch.int32))
    pat = torch.zeros(N, dtype=torch.int32).index_add_(0, rep, npatch)
    alive = cnt > 0
    den = pat[alive] if weighted else cnt[alive]
    return acc[alive] / den.unsqueeze(-1)


def aggregate_groups(x, npatch, tlbr, rep, weighted_avg):
    """quadtree_temporal_merger.py:123-171: sums in the INPUT dtype, ascending node order; the
    survivor keeps its own box (quirk Q3)."""
    N = x.shape[0]
    # keep the int32 index: for bf16/fp16 rows ATen's int32-index CPU path adds row by row in the input
    # dtype (one rounding per add, quirk Q8), while the int64-index path accumulates in fp32.
    rep = rep.to(torch.int32)
    acc = torch.zeros_like(x).index_add_(0, rep, x)
    cnt = torch.zeros(N, dtype=torch.int32).index_add_(0, rep, torch.ones(N, dtype=torch.int32))
    pat = torch.zeros(N, dtype=torch.int32).index_add_(0, rep, npatch)
    alive = cnt > 0
    pat = pat[alive]
    den = pat if weighted_avg else cnt[alive]
    return acc[alive] / den.unsqueeze(-1), pat, tlbr[alive]


def temporal_merge(x, tlbr, npatch, temporal_thresh, weighted_avg=False, head_dim=None, slow_ver=False,
                   return_debug=False):
    """cross_frame_node_merging_fast / _slow (quadtree_temporal_merger.py:271-301)."""
    if slow_ver:
        kept = slow_pairs(x, tlbr, temporal_thresh)
        cand = None
    else:
        cand = candidate_pairs(tlbr)
        kept = keep_similar_pairs(x, cand, temporal_thresh, head_dim)
    rep, iters = propagate_labels(kept, x.shape[0])
    out = aggregate_groups(x, npatch, tlbr, rep, weighted_avg)
    if return_debug:
        return out, dict(candidates=cand, kept=kept, rep=rep, iters=iters)
    return out


# ----------------------------------------------------------------------------------------------
# L1 function boundary (same signatures as the reference)
# ----------------------------------------------------------------------------------------------

def get_quadtree_features(_video_feature, threshold, temporal_thresh=-1.0, root_level=0, weighted_avg=False,
                          vis_flag=False, slow_ver=False, head_dim=None, pos_embs=None,
                          pos_emb_weighted_avg=False, return_debug=False):
    """quadtree_interface.py:5-13 -> quadtree_builder.py:85-235.
    Input logical [T, C, H, W]; returns (features [N', C] input dtype, num_patches [N'] int32,
    tlbr [N', 5] int32)."""
    if vis_flag:
        raise NotImplementedError("oracle covers the non-vis path only")
    x = _video_feature.permute(0, 2, 3, 1)                          # [T, H, W, C] (a view for production input)
    T, H, W, C = x.shape
    geom = Geometry(H, W, root_level)
    mode = "sum" if weighted_avg else "avg"
    dbg = {}
    pos = None
    if pos_embs is not None:
        # pos-emb pooling exists for even or both-odd levels only (quadtree_spatial_merger.py:88-153)
        for lvl in range(1, geom.n_level):
            h, w = geom.dims[lvl]
            if (h % 2) != (w % 2):
                raise RuntimeError("position-embedding pooling is undefined for mixed-parity grids "
                                   f"({h}x{w}); the reference fails here too")
        pmode = "sum" if pos_emb_weighted_avg else "avg"
        pos = [p.permute(0, 2, 3, 1) for p in pos_embs]               # cos, sin as [T, H, W, Cp]
    if geom.n_level == 1:                                           # no pyramid (:146-174)
        feats = x.reshape(T * H * W, C)
        t = torch.arange(T, dtype=torch.int32).repeat_interleave(H * W)
        tlbr = torch.cat([t[:, None], geom.boxes(0).repeat(T, 1)], dim=1)
        pos_nodes = [p.reshape(T * H * W, -1) for p in pos] if pos is not None else None
    else:
        maps = build_pyramid(x, geom, mode)
        if pos is not None:
            pmaps = [build_pyramid(p, geom, pmode) for p in pos]
            feats, tlbr, pos_nodes = spatial_split(maps, geom, threshold, head_dim, extra_maps=pmaps)
        else:
            feats, tlbr = spatial_split(maps, geom, threshold, head_dim)
            pos_nodes = None
    npatch = (tlbr[:, 3] - tlbr[:, 1]) * (tlbr[:, 4] - tlbr[:, 2])
    dbg["spatial_tlbr"] = tlbr
    pos_out = None
    if temporal_thresh > 0:
        node_patch = npatch
        res = temporal_merge(feats, tlbr, npatch, temporal_thresh, weighted_avg, head_dim, slow_ver,
                             return_debug=True)
        (feats, npatch, tlbr), d2 = res
        dbg.update(d2)
        if pos_nodes is not None:
            pos_out = tuple(aggregate_extra(p, d2["rep"], node_patch, pos_emb_weighted_avg) for p in pos_nodes)
    else:
        if weighted_avg:
            feats = feats / npatch.unsqueeze(1)                      # :225-226
        if pos_nodes is not None:
            if not pos_emb_weighted_avg:
                # quirk Q11: `pos_embs_cos` is only assigned on the temporal or the weighted path (:228-233)
                raise UnboundLocalError("local variable 'pos_embs_cos' referenced before assignment (the reference leaves "
                                        "it unset when pos_embs is given with temporal_thresh <= 0 and "
                                        "pos_emb_weighted_avg=False)")
            pos_out = tuple(p / npatch.unsqueeze(1) for p in pos_nodes)
    if pos_out is not None:
        if return_debug:
            return feats, npatch, tlbr, pos_out, dbg
        return feats, npatch, tlbr, pos_out
    if return_debug:
        return feats, npatch, tlbr, dbg
    return feats, npatch, tlbr


# ----------------------------------------------------------------------------------------------
# ToMe baseline                                          (tome_token_merger.py)
# ----------------------------------------------------------------------------------------------

def tome_match(metric, r):
    """bipartite_soft_matching (tome_token_merger.py:13-41) on one [n, D] metric.
    Returns (unm [na-r], src [r], dst [r]) as indices into the even (a) / odd (b) halves."""
    m = metric / metric.norm(dim=-1, keepdim=True)                  # no eps (:32)
    a, b = m[0::2], m[1::2]
    scores = a @ b.transpose(-1, -2)
    best, arg = scores.max(dim=-1)
    order = best.argsort(dim=-1, descending=True)
    src, unm = order[:r], order[r:]
    return unm, src, arg[src]


def tome_video(x_tchw, prune_ratio, n_head=1):
    """tome_per_video (tome_token_merger.py:133-152) incl. merge / merge_wavg (:43-57, :77-91)."""
    T, C, H, W = x_tchw.shape
    x = x_tchw.permute(0, 2, 3, 1).reshape(T * H * W, C)
    n = x.shape[0]
    target = math.ceil(n * (1 - prune_ratio))
    idx = torch.arange(n)
    size = None
    first = True
    while first or x.shape[0] > target:
        first = False
        n = x.shape[0]
        r = min(n - target, n // 2)
        if size is None:
            size = torch.ones_like(x[:, :1])
        if r <= 0:
            # the reference's do_nothing(x, mode=None) stand-in (:9-10, :28-29) is then called as
            # merge(x*size, token_idx, mode="sum") by merge_wavg (:87) and raises TypeError
            raise TypeError("do_nothing() got multiple values for argument 'mode' "
                            "(prune_ratio <= 0 is unusable in the reference)")
        metric = x.reshape(n, n_head, C // n_head).mean(1)
        unm, src, dst = tome_match(metric, r)

        def merge(v):
            va, vb = v[0::2], v[1::2]
            vb = vb.scatter_add(0, dst[:, None].expand(-1, v.shape[1]), va[src])
            return torch.cat([va[unm], vb], dim=0)

        xs = merge(x * size)
        size = merge(size)
        x = xs / size
        idx = torch.cat([idx[0::2][unm], idx[1::2]])
    return x, idx


def get_tome_features(_video_feature, prune_ratio, tome_ver, n_head=1):
    """tome_interface.py:3-9.  'frame' is broken upstream for T > 1 (quirk Q4): the token-index tensor
    has batch 1 while the features have batch T, and torch.gather raises RuntimeError."""
    if tome_ver == "frame":
        if _video_feature.shape[0] > 1:
            raise RuntimeError("tome_ver='frame' is broken in the reference for T > 1 "
                               "(token_idx batch 1 vs T, tome_token_merger.py:52-54)")
        return tome_video(_video_feature, prune_ratio, n_head)
    if tome_ver == "video":
        return tome_video(_video_feature, prune_ratio, n_head)
    return None                                                      # 'snippet' stub and unknown versions
