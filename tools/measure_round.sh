#!/bin/bash
# Run on the GPU box (via gpurun): quick bench lines for a set of library switches + one rocprofv3 kernel trace.
# usage: tools/measure_round.sh <tag> ["ENV=1 ENV2=2" ...]   (each quoted argument = one extra bench variant)
set -u
TAG=${1:-m}; shift || true
mkdir -p gpurun_out
B="python bench.py --steps 3 --warmup 1 --videos-per-step 512 --no-cpu-baseline --profile-calls 256"
$B > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
for V in "$@"; do
    N=$(echo "$V" | tr ' =' '__')
    env $V $B --no-extensions > gpurun_out/${TAG}_bench_$N.json 2> gpurun_out/${TAG}_bench_$N.err
done
bash tools/profile_bench.sh ${TAG}_prof --steps 1 --warmup 1 --videos-per-step 512 --profile-calls 8
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")) :
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f.split("/")[-1], "value", d["value"], "wall_ms", r["wall_ms_per_video"], "frac", r["frac"], r["kernel_ms"],
          {k: d[k]["value"] for k in ("batched_extension", "threaded_dropin_extension", "tome_extension") if k in d})
PY
cat gpurun_out/${TAG}_prof_kernels.md
