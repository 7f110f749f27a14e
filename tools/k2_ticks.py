"""Life of single workgroups of the pair kernel (wall_clock64 stamps, 100 MHz).  STTM_K2_TICKS=1 makes the library stamp."""
import ctypes, os, sys
os.environ["STTM_K2_TICKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
T, C, H, W = 128, 1024, 14, 14
pool = [synth_video(T, C, H, W, seed=i, device=dev, gen_device=dev) for i in range(4)]
nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, 0, 1)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
N = T * H * W
feat = torch.empty((N, C), device=dev); npatch = torch.empty(N, dtype=torch.int32, device=dev)
tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev); counts = torch.zeros(8, dtype=torch.int32, device=dev)
lib.sttm_debug_colscratch_offset.restype = ctypes.c_size_t
lib.sttm_debug_colscratch_offset.argtypes = [ctypes.c_int] * 6
off = lib.sttm_debug_colscratch_offset(T, H, W, C, 0, 1) // 8 + 64
acc = None; nc = 0
for it in range(12):
    x = pool[it % 4]
    rc = lib.sttm_quadtree_merge(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, 0,
                                 0.85, 0.55, 1, 0, 0, 0, ws.data_ptr(), nbytes, feat.data_ptr(), npatch.data_ptr(),
                                 tlbr.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    tk = ws.view(torch.int64)[off:off + 9].cpu().tolist()
    if it >= 2:
        d = [(tk[0] - tk[8]) / 100.0] + [(b - a) / 100.0 for a, b in zip(tk[0:3], tk[1:4])]
        acc = d if acc is None else [p + q for p, q in zip(acc, d)]
        nc += tk[4]
names = ["start after workgroup 0", "lists + box tests", "candidate dot products", "tail"]
print("pair kernel, workgroup", os.environ.get("STTM_K2_TICKS_WG", "0"), f"({nc / 10:.1f} candidates) (us, mean of 10): " +
      ", ".join(f"{n} {v / 10:.2f}" for n, v in zip(names, acc)) + f"; life {sum(acc[1:]) / 10:.2f}")
