#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of tools/micro/floor_probe -> gpurun_out/<tag>_floor.md
TAG=${1:-floor}
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp && rm -rf /tmp/fp_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fp_$TAG -o fp -- "$REPO/tools/micro/floor_probe" > "$REPO/gpurun_out/${TAG}_floor.log" 2>&1
cd "$REPO"
DB=$(find /tmp/fp_$TAG -name '*.db' | head -1)
python tools/prof_summary.py "$DB" "gpurun_out/${TAG}_floor.md"
cat "gpurun_out/${TAG}_floor.md"
