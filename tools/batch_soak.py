"""Soak of the stage-skewed batch entry point (GPU box): many calls, EVERY output of every call compared bit for bit with the one-video
call's, from one host thread and from two host threads on their own streams; mixed call sizes; prints mismatches (expected: none)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import get_quadtree_features, get_quadtree_features_batch
from sttm_amd.synth import synth_video
dev = torch.device("cuda:0")
CALLS = int(os.environ.get("CALLS", "200"))
pool = [synth_video(128, 1024, 14, 14, seed=4000 + i, device=dev, gen_device=dev) for i in range(8)]
pool += [synth_video(64, 1024, 14, 14, seed=4100 + i, device=dev, gen_device=dev) for i in range(4)]        # a second shape in the same call
ref = [get_quadtree_features(v, 0.85, 0.55, 1) for v in pool]
ref = [tuple(t.clone() for t in r) for r in ref]
torch.cuda.synchronize()
bad, done = [], [0]
lock = threading.Lock()


def work(tid, calls):
    g = torch.Generator().manual_seed(tid)
    st = torch.cuda.Stream(device=dev) if tid else torch.cuda.current_stream(dev)
    with torch.cuda.stream(st):
        for c in range(calls):
            n = int(torch.randint(1, 97, (1,), generator=g))
            ids = torch.randint(0, len(pool), (n,), generator=g).tolist()
            outs = get_quadtree_features_batch([pool[i] for i in ids], 0.85, 0.55, 1)
            for i, o in zip(ids, outs):
                ok = all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(o, ref[i]))
                if not ok:
                    with lock:
                        bad.append((tid, c, i, o[0].shape[0], ref[i][0].shape[0]))
            with lock:
                done[0] += n
    st.synchronize()


t0 = time.perf_counter()
work(0, CALLS)
print(f"one thread: {done[0]} videos in {CALLS} calls, {len(bad)} mismatches, {time.perf_counter() - t0:.1f} s", flush=True)
done[0] = 0
th = [threading.Thread(target=work, args=(k, CALLS // 2)) for k in (1, 2)]
t0 = time.perf_counter()
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
print(f"two threads on own streams: {done[0]} videos, {len(bad)} mismatches in total, {time.perf_counter() - t0:.1f} s")
print("mismatches:", bad[:10])
sys.exit(1 if bad else 0)
