"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's DyCoke stage-1 pruning
(`dycoke_ttm`, token_merging_utils/dycoke_merger.py:8-83).

Parity pinned: checked against tests/golden/dyc_*.npz, produced by tests/golden/make_golden_dycoke.py from the reference's
own function (indices and features bit-exact on CPU).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product path (sttm_amd/) never does.

Structure (own, not the reference's two loops): a table says for every frame which earlier frame prunes it, then each
frame is emitted once.
    frame f odd                      -> pruned against f-1   (pass 1, :13-45)
    frame f = 4j+2 with 4j < T-4     -> pruned against f-2   (pass 2, :54-79; replaces the whole copy pass 1 stored)
    every other frame                -> kept whole            (incl. an odd last frame, :47-52)
A pruned frame keeps its k = int((1 - prune_ratio) * P) tokens with the SMALLEST cosine similarity to the same position of
its partner, in ascending-similarity order (`topk(largest=False)`).  Cosine = sum((a / max(|a|, 1e-8)) * (b / max(|b|, 1e-8)))
(F.cosine_similarity's normalise-first form).  T < 5 makes the reference stack an empty list -> RuntimeError.
"""
import torch


def token_cosine(a, b, eps=1e-8):
    na = torch.linalg.vector_norm(a, 2, dim=1, keepdim=True).clamp_min(eps)
    nb = torch.linalg.vector_norm(b, 2, dim=1, keepdim=True).clamp_min(eps)
    return ((a / na) * (b / nb)).sum(1)


def pruning_partner(f, T):
    if f % 2 == 1:
        return f - 1
    if f % 4 == 2 and f - 2 < T - 4:
        return f - 2
    return None


def dycoke_ttm(image_feature, num_frames, prune_ratio=0.7, return_sims=False):
    T = int(num_frames)
    P = image_feature.shape[0] // T
    if T < 5:
        raise RuntimeError("stack expects a non-empty TensorList")
    k = int((1 - prune_ratio) * P)
    frames = image_feature[:T * P].reshape(T, P, -1)
    feats, ids, sims = [], [], {}
    for f in range(T):
        partner = pruning_partner(f, T)
        if partner is None:
            keep = torch.arange(P)
        else:
            s = token_cosine(frames[partner], frames[f])
            sims[f] = s
            keep = s.topk(k, largest=False).indices
        feats.append(frames[f][keep])
        ids.append(keep + f * P)
    out = (torch.cat(feats, 0), torch.cat(ids, 0))
    return out + (sims,) if return_sims else out
