"""Stage-skewed batch entry point (sttm_quadtree_merge_batch, round 5): videos/s against the number of internal streams,
the videos per launch set on a stream and the videos per call; outputs checked bit-identical to one-video calls first.
Run on the GPU box:  python tools/batch_pipeline.py [quick]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib, get_quadtree_features, get_quadtree_features_batch
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
lib = _lib.load()
T, C = int(os.environ.get("T", "128")), int(os.environ.get("C", "1024"))
dt = torch.float32 if os.environ.get("DT", "f32") == "f32" else torch.bfloat16
P = 8
pool = [synth_video(T, C, 14, 14, seed=100 + i, dtype=dt, device=dev, gen_device=dev) for i in range(P)]
ref = [get_quadtree_features(v, 0.85, 0.55, 1) for v in pool]
torch.cuda.synchronize()


def cfg(key, val):
    assert lib.sttm_configure(key.encode(), int(val)) == 0, key


def check(streams, sub):
    cfg("batch_streams", streams); cfg("batch_sub", sub)
    out = get_quadtree_features_batch(pool + pool[:5], 0.85, 0.55, 1)
    torch.cuda.synchronize()
    for j, (f, n, t) in enumerate(out):
        rf, rn, rt = ref[j % P]
        assert torch.equal(t, rt) and torch.equal(n, rn) and torch.equal(f, rf), (streams, sub, j)


def rate(fn, n):
    fn(); torch.cuda.synchronize()
    best = 0.0
    for _ in range(2):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t0))
    return best


N = 1536
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
print("one video per call, one stream: %.0f videos/s" % rate(lambda: [get_quadtree_features(pool[i % P], 0.85, 0.55, 1) for i in range(N)], N), flush=True)


def threads(nth):
    def work(k):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            for i in range(k, N, nth):
                get_quadtree_features(pool[i % P], 0.85, 0.55, 1)
        st.synchronize()
    th = [threading.Thread(target=work, args=(k,)) for k in range(nth)]
    [t.start() for t in th]; [t.join() for t in th]


for nth in (3, 4):
    print(f"{nth} host threads, one stream each: {rate(lambda: threads(nth), N):.0f} videos/s", flush=True)

combos = [(0, 16, 16), (0, 16, 48), (3, 1, 48), (2, 4, 48), (3, 4, 48), (4, 4, 48), (2, 8, 48), (3, 8, 48), (3, 3, 48), (3, 4, 96), (4, 4, 96), (2, 6, 96), (3, 6, 96), (2, 12, 96), (3, 16, 96)]
if quick:
    combos = [(0, 16, 16), (3, 1, 48), (4, 4, 48), (3, 8, 48), (4, 4, 96)]
for streams, sub, B in combos:
    check(streams, sub)

    def f(B=B):
        for b0 in range(0, N, B):
            get_quadtree_features_batch([pool[(b0 + k) % P] for k in range(B)], 0.85, 0.55, 1)
    tag = "lockstep (16 videos per launch set)" if streams == 0 else f"{streams} internal streams x {sub} video(s) per launch set"
    print(f"batch call of {B:3d} videos, {tag}: {rate(f, N):.0f} videos/s", flush=True)
