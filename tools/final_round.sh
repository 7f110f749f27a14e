#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/ records for one build -- the PMC passes (HBM traffic, tagged with the build), the full
# default bench line (batch mode), rocprofv3 kernel stats of a short one-call-per-video run (the kernels alone) and of a short batch-mode run
# (+ the device's busy fraction), the launch timeline, bench under torchrun with one rank, the batch soak, the GPU test summary.
# usage: tools/final_round.sh <tag>
set -u
TAG=${1:-r05z}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.txt 2>&1; tail -2 gpurun_out/${TAG}_tests.txt
# the PMC passes come first: they write profiles/pmc_traffic.json for THIS build tag, which the bench lines then carry as roofline.traffic
bash tools/pmc_passes.sh ${TAG} > /dev/null 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash tools/profile_bench.sh ${TAG}_dropin --mode dropin --steps 2 --warmup 1 --videos-per-step 512 --profile-calls 8
bash tools/prof_batch_busy.sh ${TAG}
export TMPDIR=/tmp
DB=$(find /tmp/prof_${TAG}_dropin -name '*.db' | head -1); python tools/prof_timeline.py "$DB" 8 gpurun_out/${TAG}_timeline.md > /dev/null 2>&1
B="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extensions --no-configs --profile-calls 64"
python $B > gpurun_out/${TAG}_plain.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 $B --gpus 1 > gpurun_out/${TAG}_torchrun.json 2>/dev/null
CALLS=400 timeout 600 python tools/batch_soak.py > gpurun_out/${TAG}_batch_soak.txt 2>&1
python - <<PY
import json
for f in ("bench", "plain", "torchrun"):
    try:
        d = json.loads(open("gpurun_out/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], "wall", r["wall_ms_per_video"], "frac", r["frac"], "traffic", r["traffic"], r["dominant_kernel"], r["kernel_ms"],
              {k: d[k]["value"] for k in ("dropin_one_call_per_video", "threaded_dropin_extension", "tome_extension") if k in d}, d.get("cpu_baseline", None) and d["cpu_baseline"]["value"])
    except Exception as e:
        print(f, "unreadable", e)
PY
head -8 gpurun_out/${TAG}_dropin_kernels.md
cat gpurun_out/${TAG}_batch_busy.md gpurun_out/${TAG}_pmc_traffic.md
tail -3 gpurun_out/${TAG}_batch_soak.txt
