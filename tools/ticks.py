"""Phase timelines of single workgroups (wall_clock64 stamps, 100 MHz) -- DEVELOPMENT build only.

    python -m sttm_amd.build --dev && STTM_LIB=dev python tools/ticks.py [k1_wg] [k2_wg] [label_column]

Prints the life of one spatial workgroup, one pair workgroup and one column's label stage (folded into the pair kernel or
stand-alone, whichever the library chose) on the headline workload.  The product library has none of these hooks."""
import ctypes, os, sys
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
lib.sttm_dev_hooks.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
T, C, H, W = int(os.environ.get("T", "128")), 1024, int(os.environ.get("H", "14")), int(os.environ.get("W", "14"))
THR, TTHR = float(os.environ.get("THR", "0.85")), float(os.environ.get("TTHR", "0.55"))
RL = int(os.environ.get("RL", "1"))          # root_level (36 x 64 tokens at RL=0: a 6-level tree, the split spatial stage)
k1_wg, k2_wg, col = (int(a) for a in (sys.argv[1:4] + ["0", "0", "0"])[:3])
x = synth_video(T, C, H, W, seed=1, device=dev, gen_device=dev)
nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, 0, RL)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
N = T * H * W
feat = torch.empty((N, C), device=dev); npatch = torch.empty(N, dtype=torch.int32, device=dev)
tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev); counts = torch.zeros(8, dtype=torch.int32, device=dev)
ticks = torch.zeros(48, dtype=torch.int64, device=dev)
lib.sttm_dev_hooks(0, ticks.data_ptr(), k1_wg, k2_wg, col)
names = {0: ("spatial workgroup", ["start", "loads+pool", "stats", "decide+emit", "stores"]),
         8: ("upper pass of the split spatial stage (trees of 4+ levels)", ["start", "prefetch+clear+aliases", "statistics", "tests", "emission walk", "stores"]),
         16: ("pair workgroup", ["start", "lists+box tests", "dots", "published"]),
         32: ("label stage", ["start", "edges+bits", "compact ids", "probe", "grid barrier", "K+replay", "sizes+results", "frame counts", "arrival"])}
acc = {k: None for k in names}
runs = 0
for it in range(12):
    ticks.zero_()
    rc = lib.sttm_quadtree_merge(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, 0,
                                 THR, TTHR, RL, 0, 0, 0, ws.data_ptr(), nbytes, feat.data_ptr(), npatch.data_ptr(),
                                 tlbr.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    tk = ticks.cpu().tolist()
    last_counts = counts.cpu().tolist()
    if it < 2:
        continue
    runs += 1
    for base, (_, nm) in names.items():
        v = tk[base:base + len(nm)]
        d = [(b - a) / 100.0 if (a and b) else 0.0 for a, b in zip(v[:-1], v[1:])]
        acc[base] = d if acc[base] is None else [p + q for p, q in zip(acc[base], d)]
    rel = [(tk[16] - tk[0]) / 100.0, (tk[32] - tk[16]) / 100.0]
for base, (title, nm) in names.items():
    print(f"{title} (us, mean of {runs}):")
    for n_, v in zip(nm[1:], acc[base]):
        print(f"  {n_:16s} {v / runs:6.2f}")
    print(f"  {'total':16s} {sum(acc[base]) / runs:6.2f}")

print("counts (nodes, candidates, edges, out, iters, overflow, leafnodes):", last_counts[:7])
