"""Phase timeline of the fused label kernel (workgroup 0), from wall_clock64 stamps (100 MHz)."""
import ctypes, os, sys
os.environ["STTM_LABEL_TICKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
T, C, H, W = 128, 1024, 14, 14
x = synth_video(T, C, H, W, seed=1, device=dev, gen_device=dev)
nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, 0, 1)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
N = T * H * W
feat = torch.empty((N, C), device=dev); npatch = torch.empty(N, dtype=torch.int32, device=dev)
tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev); counts = torch.zeros(8, dtype=torch.int32, device=dev)
lib.sttm_debug_colscratch_offset.restype = ctypes.c_size_t
lib.sttm_debug_colscratch_offset.argtypes = [ctypes.c_int] * 6
off = lib.sttm_debug_colscratch_offset(T, H, W, C, 0, 1)
assert off % 8 == 0 and ws.data_ptr() % 8 == 0
names = ["start", "gathered", "probed", "barrier1", "replayed", "counted", "offsets", "frame_cnt", "filled+sorted", "counters",
         "ticket", "prefix+publish (last column only)"]
acc = None
for it in range(12):
    rc = lib.sttm_quadtree_merge(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, 0,
                                 0.85, 0.55, 1, 0, 0, 0, ws.data_ptr(), nbytes, feat.data_ptr(), npatch.data_ptr(),
                                 tlbr.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    pos = off // 8
    tk = ws.view(torch.int64)[pos:pos + 14].cpu().tolist()
    if it == 11: print('   shader clock during the kernel: %.0f MHz (clock64 delta / wall_clock64 delta)' % ((tk[13] - tk[12]) / ((tk[11] - tk[0]) / 100.0)))
    tk = tk[:12]
    if it >= 2:
        d = [(b - a) / 100.0 for a, b in zip(tk[:-1], tk[1:])]
        acc = d if acc is None else [p + q for p, q in zip(acc, d)]
n = 10
print("fused label kernel, workgroup", os.environ.get("STTM_LABEL_TICKS_WG", "0"), "mean of", n, "runs (us):")
for nm, v in zip(names[1:], acc):
    print(f"  {nm:14s} {v / n:6.2f}")
print(f"  total          {sum(acc) / n:6.2f}")
