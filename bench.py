#!/usr/bin/env python3
"""Benchmark of the STTM merge hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): synth-v1 videos of 128 frames x 14x14 tokens x 1024 channels, fp32, in the
production layout (channels-last view), full STTM = quadtree spatial merge (thr 0.85, root_level 1) +
temporal merge (thr 0.55) -- the LLaVA-Video-7B / Video-MME "50 % budget" preset of the reference
(scripts/eval/run_vidqa.sh:58).  A step = `--videos-per-step` (32) videos through get_quadtree_features, one
after the other (the reference API is batch-1), inputs resident in HBM, outputs (incl. the host-visible
token count) produced.  Videos are independent, so N GPUs each take their own videos (weak scaling);
the only collective is the final all-gather of the per-video token counts over RCCL.

Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement" for every field).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
KERNELS = ["quadtree_spatial", "temporal_pairs", "labels_scan", "group_mean"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--videos-per-step", type=int, default=32)
    ap.add_argument("--pool", type=int, default=8, help="distinct videos resident per GPU (> L3 capacity in total)")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the batched-extension leg (e.g. under rocprofv3)")
    return ap.parse_args()


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):      # under torchrun (also with one rank)
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from sttm_amd import _lib
    from sttm_amd.quadtree_interface import get_quadtree_features, quadtree_merge_raw
    from sttm_amd.synth import synth_video
    lib = _lib.load()

    T, C, H, W = args.frames, 1024, 14, 14
    thr, tthr, root = 0.85, 0.55, 1
    V, P, K, Wm = args.videos_per_step, max(1, args.pool), args.steps, args.warmup

    # ---- inputs: P distinct videos per GPU, generated on the device (same distribution as the CPU stream) ----
    pool = [synth_video(T, C, H, W, seed=100000 * (rank + 1) + i, device=dev, gen_device=dev) for i in range(P)]
    torch.cuda.synchronize()
    log(f"pool of {P} videos ready")

    def run_step(s, sink):
        for v in range(V):
            x = pool[(s * V + v) % P]
            feat, npatch, tlbr = get_quadtree_features(x, thr, tthr, root)
            sink.append(feat.shape[0])

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    sink = []
    for s in range(Wm):
        run_step(s, sink)
    if dist is not None:
        # warm the collectives the timed region uses (RCCL builds its communicator / channels lazily, ~50 ms once)
        from sttm_amd.distributed import gather_counts, shard_videos
        ids_w = shard_videos(world * K * V, world, rank)
        cw = gather_counts(ids_w, [1] * len(ids_w), world * K * V, dev, dist)
        assert int((cw > 0).sum()) == world * K * V          # also loads the torch kernels the timed check uses
        tw = torch.zeros(1, dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        barrier()
    torch.cuda.synchronize()
    sink = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        run_step(s, sink)
    t_issue = time.perf_counter() - t0
    if dist is not None:      # the one exchange step of the job: per-video merged-token counts to every rank
        from sttm_amd.distributed import gather_counts, shard_videos
        ids = shard_videos(world * K * V, world, rank)
        all_counts = gather_counts(ids, sink, world * K * V, dev, dist)
        t_g = time.perf_counter() - t0
        n_seen = int((all_counts > 0).sum())
        assert n_seen == world * K * V, f"gather saw {n_seen} of {world * K * V} videos"
        t_a = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_s = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        log(f"  breakdown: issue {t_issue*1e3:.2f} gather-issued {t_g*1e3:.2f} checked {t_a*1e3:.2f} synced {t_s*1e3:.2f} barrier {elapsed*1e3:.2f} ms")
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    videos = world * K * V
    value = videos / elapsed
    log(f"timed region: {videos} videos in {elapsed:.4f} s = {value:.1f} videos/s (merge calls returned after {t_issue:.4f} s)")

    # ---- extension leg: the same K steps through the batched API (videos of a step issued on two side streams, so the
    #      latency-bound label kernel of one video overlaps the bandwidth-bound kernels of the next) ----------------------
    from sttm_amd.quadtree_interface import get_quadtree_features_batch
    def run_step_batched(s):
        vids = [pool[(s * V + v) % P] for v in range(V)]
        return get_quadtree_features_batch(vids, thr, tthr, root, n_streams=int(os.environ.get("STTM_BATCH_STREAMS", "3")))
    batched_value = None
    if not args.no_batched:
        run_step_batched(0)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for s in range(K):
            run_step_batched(s)
        torch.cuda.synchronize()
        barrier()
        tb = time.perf_counter() - tb0
        if dist is not None:
            tmaxb = torch.tensor([tb], dtype=torch.float64, device=dev)
            dist.all_reduce(tmaxb, op=dist.ReduceOp.MAX)
            tb = float(tmaxb.item())
        batched_value = videos / tb
        log(f"batched extension: {videos} videos in {tb:.4f} s = {batched_value:.1f} videos/s")

    # ---- extension leg 2: the SAME one-video-per-call API from three host threads, each on its own stream (what a serving
    #      process with several request threads does); skipped together with the batched leg ---------------------------------
    threaded_value = None
    if not args.no_batched:
        import threading
        NTH = 3
        def worker(k, n_steps):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for s in range(n_steps):
                    for v in range(k, V, NTH):
                        get_quadtree_features(pool[(s * V + v) % P], thr, tthr, root)
            st.synchronize()
        def run_threads(n_steps):
            th = [threading.Thread(target=worker, args=(k, n_steps)) for k in range(NTH)]
            [t.start() for t in th]
            [t.join() for t in th]
        run_threads(1)
        torch.cuda.synchronize()
        barrier()
        tt0 = time.perf_counter()
        run_threads(K)
        torch.cuda.synchronize()
        barrier()
        tt = time.perf_counter() - tt0
        if dist is not None:
            tmaxt = torch.tensor([tt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmaxt, op=dist.ReduceOp.MAX)
            tt = float(tmaxt.item())
        threaded_value = videos / tt
        log(f"threaded drop-in extension: {videos} videos in {tt:.4f} s = {threaded_value:.1f} videos/s")

    # ---- roofline leg: the same K steps again with HIP events around every kernel of every call ----------
    lib.sttm_profile_enable(1)
    ms = (ctypes.c_float * 4)()
    tot = [0.0] * 4
    nodes = merged = leafnodes = 0
    calls = 0
    for s in range(K):
        for v in range(V):
            x = pool[(s * V + v) % P]
            _, _, _, cnt = quadtree_merge_raw(x, thr, tthr, root, False, None)
            _lib.raise_for(lib.sttm_profile_last(ms))
            for i in range(4):
                tot[i] += ms[i]
            nodes += cnt[_lib.CNT_NODES]
            leafnodes += cnt[_lib.CNT_LEAFNODES]
            merged += cnt[_lib.CNT_OUT]
            calls += 1
    lib.sttm_profile_enable(0)
    log("roofline leg done: " + ", ".join(f"{k}={t / calls:.4f} ms" for k, t in zip(KERNELS, tot)))
    avg_ms = [t / calls for t in tot]
    n_avg, m_avg, l_avg = nodes / calls, merged / calls, leafnodes / calls
    es = 4
    thw = T * H * W
    kernel_bytes = [                                     # algorithmic (compulsory) bytes of each kernel per launch
        es * C * thw + es * C * (n_avg - l_avg) + 16 * thw,   # read every token once, write every POOLED node once (1x1
                                                              # nodes stay in x), meta + inverse norm + node list per token
        es * C * n_avg,                                   # every node row read once (pairs share rows)
        4 * thw * 6,                                      # label / group / rank tables, once each
        es * C * n_avg + es * C * m_avg,                  # read node rows, write merged rows
    ]
    # the roofline is quoted for the dominant HBM-bound kernel; the label kernel (index 2) moves < 1 % of the bytes and is
    # latency-bound by construction (16 workgroups), it is reported in kernel_ms but never as the roofline kernel
    dom = max((0, 1, 3), key=lambda i: avg_ms[i])
    pipeline_bytes = es * C * thw + es * C * m_avg + 24 * m_avg      # B per video
    dom_gbs = kernel_bytes[dom] / (avg_ms[dom] * 1e-3) / 1e9
    pipe_gbs = pipeline_bytes / (sum(avg_ms) * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            rec = json.load(open(pmc_path))
            if rec.get("workload") == f"T{T}_14x14x1024_f32_sttm_0.85_0.55":
                traffic = rec.get("hbm_bytes_per_launch", {}).get(KERNELS[dom])
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": KERNELS[dom],
        "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dom_gbs / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "kernel_ms": {k: round(v, 4) for k, v in zip(KERNELS, avg_ms)},
        "kernel_algorithmic_MB": {k: round(b / 1e6, 2) for k, b in zip(KERNELS, kernel_bytes)},
        "pipeline": {"algorithmic_MB_per_video": round(pipeline_bytes / 1e6, 2), "device_ms_per_video": round(sum(avg_ms), 4),
                     "achieved": round(pipe_gbs, 1), "frac": round(pipe_gbs / HBM_PEAK_GBS, 4)},
    }

    # ---- CPU baseline leg (rank 0, N = 1 only): the oracle on a bounded sample of the same workload -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import sttm_oracle as O
        ncpu = os.cpu_count() or 1
        sample = [pool[i % P].cpu() for i in range(min(P, 8))]             # same tensors the GPU path was timed on
        # pick the thread count the CPU path likes best on this host (best case for the baseline)
        best_thr, best_t = None, None
        for nthr in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
            torch.set_num_threads(nthr)
            O.get_quadtree_features(sample[0], thr, tthr, root)               # warm-up, untimed
            c0 = time.perf_counter()
            O.get_quadtree_features(sample[0], thr, tthr, root)
            dt = time.perf_counter() - c0
            log(f"cpu baseline: {nthr:3d} threads -> {dt * 1e3:.1f} ms / video")
            if best_t is None or dt < best_t:
                best_thr, best_t = nthr, dt
        torch.set_num_threads(best_thr)
        done, match, spent = 0, 0, 0.0
        wall0 = time.perf_counter()
        while spent < args.cpu_seconds and done < 64 and time.perf_counter() - wall0 < 60.0:
            x = sample[done % len(sample)]
            c0 = time.perf_counter()
            ef, en, et = O.get_quadtree_features(x, thr, tthr, root)
            spent += time.perf_counter() - c0
            if done < len(sample):                                             # parity of the timed GPU path on the same input
                f, n, t = get_quadtree_features(pool[done % P], thr, tthr, root)
                if t.shape == et.shape and torch.equal(t.cpu(), et) and torch.equal(n.cpu(), en) \
                        and float((f.cpu() - ef).abs().max()) <= 1e-5:
                    match += 1
            done += 1
        checked = min(done, len(sample))
        log(f"cpu baseline: {done} videos in {spent:.2f} s with {best_thr} threads; {match}/{checked} index-exact")
        cpu = {"value": round(done / spent, 3), "unit": "videos/s", "cores": best_thr, "kind": "port",
               "sample": f"{done} runs over {len(sample)} distinct synth-v1 videos (the GPU pool), T={T} 14x14x1024 fp32, "
                         f"STTM(0.85,0.55,root=1), oracle/sttm_oracle.py on torch CPU, best of 8/16/32/64/128 threads = {best_thr} "
                         f"(host has {ncpu} logical CPUs), 1 warm-up",
               "index_exact_videos": match, "videos_checked": checked}

    if rank == 0:
        out = {
            "metric": "videos/sec (128-frame, 14x14x1024 tokens) STTM merge; merged-index match vs ref",
            "value": round(value, 2), "unit": "videos/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synth-v1 T={T} 14x14x1024 fp32, STTM spatial 0.85 + temporal 0.55, root_level 1",
                       "videos_per_step": V, "pool_per_gpu": P, "global_videos": videos, "parallelism": f"videos sharded x{world}",
                       "api": "get_quadtree_features, one video per call (the reference's drop-in boundary)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "batched_extension": None if batched_value is None else {"value": round(batched_value, 2), "unit": "videos/s",
                                  "api": "get_quadtree_features_batch: the step's videos in one call, 3 side streams "
                                         "(not the reference's batch-1 API; identical outputs)"},
        }
        out["threaded_dropin_extension"] = None if threaded_value is None else {
            "value": round(threaded_value, 2), "unit": "videos/s",
            "api": "get_quadtree_features, one video per call, from 3 host threads with one stream each (identical outputs)"}
        if cpu:
            out["index_match"] = cpu["index_exact_videos"] / max(1, cpu["videos_checked"])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
