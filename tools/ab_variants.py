"""A/B of library switches on one GPU: for every `key=value[,key=value]` argument, (1) outputs bit-identical to the default
configuration on a few videos, (2) per-kernel HIP-event times and wall time per video (drop-in API, pool of device-resident
videos).   usage: python tools/ab_variants.py [--shape headline|prod|both] default pairs_var=9 no_dense=1,gm_split=16 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sttm_amd import _lib, get_quadtree_features
from sttm_amd.quadtree_interface import quadtree_merge_raw
from sttm_amd.synth import synth_video

dev = torch.device("cuda:0")
SHAPES = {
    "headline": (128, 1024, 14, 14, torch.float32, 0.85, 0.55),
    "prod": (128, 3584, 14, 14, torch.bfloat16, 0.85, 0.55),
    "c4": (128, 1024, 20, 36, torch.float32, 0.85, 0.60),
    "c5": (180, 1024, 14, 14, torch.float32, 0.94, 0.82),
    "c2": (64, 1024, 14, 14, torch.float32, 0.85, 0.65),
    "c4b": (128, 1024, 18, 26, torch.float32, 0.85, 0.60),
    "c4c": (128, 1024, 13, 24, torch.float32, 0.85, 0.60),
    "t256": (256, 1024, 14, 14, torch.float32, 0.85, 0.55),
    "wide": (128, 8192, 14, 14, torch.bfloat16, 0.85, 0.55),
    "bf16n": (128, 2048, 14, 14, torch.bfloat16, 0.85, 0.55),
    # deep trees (an 8th entry = root_level): 36 x 64 tokens at root_level 0 / 1 = 6- / 5-level trees, 27 x 27 at 0 = 5 levels
    "deep6": (16, 1024, 36, 64, torch.float32, 0.85, 0.55, 0),
    "deep5": (16, 1024, 36, 64, torch.float32, 0.85, 0.55, 1),
    "deep5b": (64, 1024, 27, 27, torch.float32, 0.85, 0.55, 0),
    "deep6h": (16, 3584, 36, 64, torch.bfloat16, 0.85, 0.55, 0),
}
KEYS_RESET = {}


def configure(spec):
    lib = _lib.load()
    for k, v in KEYS_RESET.items():
        lib.sttm_configure(k.encode(), v)
    if spec == "default":
        return
    for kv in spec.split(","):
        k, v = kv.split("=")
        KEYS_RESET.setdefault(k, 0)
        if lib.sttm_configure(k.encode(), int(v)) != 0:
            raise SystemExit(f"unknown key {k}")


def main():
    args = sys.argv[1:]
    shapes = ["headline"]
    if args and args[0] == "--shape":
        shapes = list(SHAPES) if args[1] == "all" else args[1].split(",")
        args = args[2:]
    specs = args or ["default"]
    for sh in shapes:
        T, C, H, W, dt, thr, tthr = SHAPES[sh][:7]
        RL = SHAPES[sh][7] if len(SHAPES[sh]) > 7 else 1
        P = 8 if sh not in ("c4", "c4b", "wide", "t256") else 4
        pool = [synth_video(T, C, H, W, seed=i, dtype=dt, device=dev, gen_device=dev) for i in range(P)]
        configure("default")
        ref = [get_quadtree_features(x, thr, tthr, RL) for x in pool]
        torch.cuda.synchronize()
        print(f"== {sh}: T={T} C={C} {H}x{W} {dt} ==", flush=True)
        for spec in specs:
            configure(spec)
            same = True
            for x, r in zip(pool, ref):
                f, n, t = get_quadtree_features(x, thr, tthr, RL)
                same &= bool(torch.equal(f, r[0]) and torch.equal(n, r[1]) and torch.equal(t, r[2]))
            ev = _lib.KernelEvents()
            tot = [0.0] * 4
            calls = 256
            for i in range(calls):
                quadtree_merge_raw(pool[i % P], thr, tthr, RL, False, None, events=ev, return_ctx=True)
                ms = ev.elapsed_ms()
                for k in range(4):
                    tot[k] += ms[k]
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n_it = 1024
                for i in range(n_it):
                    get_quadtree_features(pool[i % P], thr, tthr, RL)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n_it)
            print(f"{spec:40s} identical={same}  K1 {tot[0] / calls * 1e3:6.2f}  K2 {tot[1] / calls * 1e3:6.2f}  K3 {tot[2] / calls * 1e3:6.2f}  "
                  f"K5 {tot[3] / calls * 1e3:6.2f} us   wall {best * 1e6:7.2f} us/video = {1 / best:8.0f} videos/s", flush=True)
        configure("default")


if __name__ == "__main__":
    main()
