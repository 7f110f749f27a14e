"""Timeline of the merged label + group-mean launch (k_labels_mean) -- DEVELOPMENT build only.

    python -m sttm_amd.build --dev && STTM_LIB=dev python tools/gm_span.py [label_column]

Stamps (wall_clock64, 100 MHz) of every group-mean workgroup (start / prefetch issued / labels seen / last wave's end) next to
the label stage's phases of one column, all relative to the first stamp of the launch."""
import ctypes, os, sys
os.environ["STTM_LIB"] = "dev"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib
from sttm_amd.synth import synth_video
lib = _lib.load()
lib.sttm_dev_hooks.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.sttm_dev_gm_span.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
T, C, H, W = int(os.environ.get("T", "128")), int(os.environ.get("C", "1024")), 14, 14
col = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dt = {"f32": torch.float32, "bf16": torch.bfloat16}[os.environ.get("DT", "f32")]
x = synth_video(T, C, H, W, seed=1, device=dev, gen_device=dev).to(dt)
code = 0 if dt == torch.float32 else 1
nbytes = lib.sttm_quadtree_workspace_bytes(T, H, W, C, code, 1)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
N = T * H * W
feat = torch.empty((N, C), device=dev, dtype=dt); npatch = torch.empty(N, dtype=torch.int32, device=dev)
tlbr = torch.empty((N, 5), dtype=torch.int32, device=dev); counts = torch.zeros(8, dtype=torch.int32, device=dev)
ticks = torch.zeros(48, dtype=torch.int64, device=dev)
nwg = 4096
span = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
lib.sttm_dev_hooks(0, ticks.data_ptr(), 0, 0, col)
lib.sttm_dev_gm_span(span.data_ptr())
import statistics as st
for it in range(6):
    ticks.zero_(); span.zero_()
    rc = lib.sttm_quadtree_merge(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), T, C, H, W, code,
                                 0.85, 0.55, 1, 0, 0, 0, ws.data_ptr(), nbytes, feat.data_ptr(), npatch.data_ptr(),
                                 tlbr.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    if it < 3:
        continue
    tk = ticks.cpu().tolist()[32:48]
    sp = span.cpu().view(-1, 8)
    sp = sp[sp[:, 0] > 0]
    t0 = min(int(sp[:, 0].min()), tk[0] if tk[0] else 1 << 62)
    lbl = [(v - t0) / 100.0 if v else None for v in tk[:9]]
    print(f"run {it}: group-mean workgroups {sp.shape[0]}; label column {col} stamps (us): " + " ".join("-" if v is None else f"{v:.1f}" for v in lbl))
    for k, name in [(0, "start"), (1, "prefetch issued"), (2, "labels seen"), (4, "w0: sizes known"), (5, "w0: gathered"), (6, "w0: stores issued"), (7, "w0: meta stored"), (3, "end (last wave)")]:
        v = sorted(((sp[:, k] - t0).float() / 100.0).tolist())
        q = lambda f: v[min(len(v) - 1, int(f * len(v)))]
        print(f"   {name:16s} min {v[0]:6.1f}  p10 {q(.1):6.1f}  p50 {q(.5):6.1f}  p90 {q(.9):6.1f}  max {v[-1]:6.1f}")
    life2 = sorted(((sp[:, 3] - sp[:, 2]).float() / 100.0).tolist())
    print(f"   after the labels: p10 {life2[len(life2)//10]:.1f}  p50 {life2[len(life2)//2]:.1f}  p90 {life2[9*len(life2)//10]:.1f}  max {life2[-1]:.1f} us")
