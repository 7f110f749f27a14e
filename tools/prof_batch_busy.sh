#!/bin/bash
# GPU box: rocprofv3 kernel trace of a short bench.py run in batch mode -> per-kernel stats + busy fraction
TAG=${1:-r05q}; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
cd /tmp && rm -rf /tmp/pbb_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pbb_$TAG -o x -- python "$REPO/bench.py" --steps 4 --warmup 2 --videos-per-step 768 --no-cpu-baseline --no-extensions --no-configs --profile-calls 8 > "$REPO/gpurun_out/${TAG}_batchprof_bench.json" 2> "$REPO/gpurun_out/${TAG}_batchprof_bench.err"
cd "$REPO"; DB=$(find /tmp/pbb_$TAG -name '*.db' | head -1)
python tools/prof_summary.py "$DB" gpurun_out/${TAG}_batch_kernels.md
python tools/prof_busy.py "$DB" gpurun_out/${TAG}_batch_busy.md
