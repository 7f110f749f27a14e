"""The drop-in call (one video per call) from several host threads, each on its own stream (GPU box): throughput."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
pool = [synth_video(128, 1024, 14, 14, seed=s, device=dev, gen_device=dev) for s in range(8)]
for v in pool: get_quadtree_features(v, 0.85, 0.55, 1)
torch.cuda.synchronize()
for nth in (1, 2, 3, 4):
    per = 320 // nth
    def work(k):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            for i in range(per):
                get_quadtree_features(pool[(k + i * nth) % 8], 0.85, 0.55, 1)
        st.synchronize()
    for rep in range(2):
        th = [threading.Thread(target=work, args=(k,)) for k in range(nth)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{nth} host thread(s), own stream each, get_quadtree_features one video per call: {per * nth / dt:.0f} videos/s")
