"""Ablation of the spatial kernel (GPU box, DEVELOPMENT build: python -m sttm_amd.build --dev): time K1 alone under
STTM_K1_ABLATE=0/1/2 (1 = stop after the statistics, 2 = loads + pooling only).
Outputs of modes 1/2 are garbage, so only the spatial kernel's time is read and nothing downstream is trusted."""
import ctypes, os, sys
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
dev = torch.device("cuda:0")
T, C, H, W = 128, int(os.environ.get("C", "1024")), 14, 14
DT = {"f32": (torch.float32, 0), "bf16": (torch.bfloat16, 1), "f16": (torch.float16, 2)}[os.environ.get("DTYPE", "f32")]
pool = [synth_video(T, C, H, W, seed=i, dtype=DT[0], device=dev, gen_device=dev) for i in range(8)]
nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, DT[1], 1)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
N = T * H * W
feat = torch.empty((N, C), device=dev, dtype=DT[0]); npatch = torch.empty(N, dtype=torch.int32, device=dev)
tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev); counts = torch.zeros(8, dtype=torch.int32, device=dev)
mode = os.environ.get("STTM_K1_ABLATE", "0")
lib.sttm_dev_hooks.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.sttm_dev_hooks(int(mode), None, 0, 0, 0)
ev = _lib.KernelEvents()
tot = 0.0; n = 0
for it in range(40):
    x = pool[it % 8]
    # temporal_thresh = -1: skips the pair kernel; garbage metadata from ablated modes is never dereferenced by it
    rc = lib.sttm_quadtree_merge_async(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, DT[1],
                                       0.85, -1.0, 1, 0, 0, 0, ws.data_ptr(), nbytes, feat.data_ptr(), npatch.data_ptr(),
                                       tlbr.data_ptr(), counts.data_ptr(), None, 0, ev.pointer(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.last_error()
    ms = ev.elapsed_ms()
    if it >= 8:
        tot += ms[0]; n += 1
print(f"C={C} {os.environ.get('DTYPE', 'f32')} STTM_K1_ABLATE={mode}: spatial kernel {tot / n * 1e3:.1f} us")
