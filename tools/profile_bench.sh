#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of a short bench.py run; summary -> gpurun_out/<tag>_kernels.md
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python "$REPO/bench.py" --no-cpu-baseline --no-extensions --no-configs "$@" \
    > "$REPO/gpurun_out/${TAG}_bench.json" 2> "$REPO/gpurun_out/${TAG}_err.log"
cd "$REPO"
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
python tools/prof_summary.py "$DB" "gpurun_out/${TAG}_kernels.md"
