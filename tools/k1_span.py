"""Occupancy curve of the spatial kernel (DEVELOPMENT build): every workgroup stamps its start and end (wall_clock64, 100 MHz).
    python -m sttm_amd.build --dev && python tools/k1_span.py
Prints how many workgroups are alive per microsecond bin, the distribution of workgroup lives and when the last ones start."""
import ctypes, os, sys
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
lib.sttm_dev_k1_span.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
T, C, H, W = int(os.environ.get("T", "128")), 1024, 14, 14
pool = [synth_video(T, C, H, W, seed=i, device=dev, gen_device=dev) for i in range(8)]
nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, 0, 1)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
N = T * H * W
feat = torch.empty((N, C), device=dev); npatch = torch.empty(N, dtype=torch.int32, device=dev)
tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev); counts = torch.zeros(8, dtype=torch.int32, device=dev)
nwg = T * 16
span = torch.zeros(2 * nwg, dtype=torch.int64, device=dev)
lib.sttm_dev_k1_span(span.data_ptr())
for it in range(6):
    x = pool[it % 8]
    rc = lib.sttm_quadtree_merge(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, 0,
                                 0.85, 0.55, 1, 0, 0, 0, ws.data_ptr(), nbytes, feat.data_ptr(), npatch.data_ptr(),
                                 tlbr.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
s = span.cpu().reshape(nwg, 2).double() / 100.0
t0 = s[:, 0].min()
st, en = s[:, 0] - t0, s[:, 1] - t0
life = en - st
print(f"workgroups {nwg}: first start 0, last start {st.max():.1f} us, last end {en.max():.1f} us")
print("life us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (life.min(), life.kthvalue(nwg // 10).values, life.median(), life.kthvalue(nwg * 9 // 10).values, life.max()))
import math
for b in range(0, int(math.ceil(float(en.max()))) + 1):
    alive = int(((st <= b + 0.5) & (en > b + 0.5)).sum())
    started = int(((st >= b) & (st < b + 1)).sum())
    ended = int(((en >= b) & (en < b + 1)).sum())
    print(f"  t={b:3d} us  alive {alive:5d}  started {started:5d}  ended {ended:5d}")
