#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/ records for one build -- the full default bench line, the rocprofv3 kernel
# stats of a short bench run, the two PMC passes (HBM traffic, tagged with the build), bench under torchrun with one rank.
# usage: tools/final_round.sh <tag>
set -u
TAG=${1:-r03z}
mkdir -p gpurun_out
# the PMC passes come first: they write profiles/pmc_traffic.json for THIS build tag, which the bench lines then carry as roofline.traffic
bash tools/pmc_passes.sh ${TAG} > /dev/null 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash tools/profile_bench.sh ${TAG}_prof --steps 2 --warmup 1 --videos-per-step 512 --profile-calls 8
B="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extensions --no-configs --profile-calls 64"
python $B > gpurun_out/${TAG}_plain.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 $B --gpus 1 > gpurun_out/${TAG}_torchrun.json 2>/dev/null
python - <<PY
import json
for f in ("bench", "plain", "torchrun"):
    try:
        d = json.loads(open("gpurun_out/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], "wall", r["wall_ms_per_video"], "frac", r["frac"], "traffic", r["traffic"], r["dominant_kernel"], r["kernel_ms"],
              {k: d[k]["value"] for k in ("batched_extension", "threaded_dropin_extension", "tome_extension") if k in d}, d.get("cpu_baseline", None) and d["cpu_baseline"]["value"])
    except Exception as e:
        print(f, "unreadable", e)
PY
head -8 gpurun_out/${TAG}_prof_kernels.md
cat gpurun_out/${TAG}_pmc_traffic.md
