"""Print the bookkeeping counters of one headline-size merge (nodes, candidates, edges, output rows, iterations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.quadtree_interface import quadtree_merge_raw
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
for root_level in (0, 1, 2):
    x = synth_video(128, 1024, 14, 14, seed=1, device=dev, gen_device=dev)
    f, n, t, cnt = quadtree_merge_raw(x, 0.85, 0.55, root_level, False, None)
    lib = _lib.load()
    print("root_level", root_level, "levels", lib.sttm_quadtree_num_levels(14, 14, root_level), "counts", cnt)
