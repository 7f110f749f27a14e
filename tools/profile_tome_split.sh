#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of tools/tome_split_probe.py -> gpurun_out/<tag>_tome_kernels.md
# usage: tools/profile_tome_split.sh <tag> [MODES]
set -u
TAG=${1:-ts}; MODES=${2:-0,1}
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
rm -rf /tmp/prof_$TAG
MODES=$MODES timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o tome -- python "$REPO/tools/tome_split_probe.py" \
    > "$REPO/gpurun_out/${TAG}_tome_probe.log" 2> "$REPO/gpurun_out/${TAG}_tome_err.log"
cd "$REPO"
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
python tools/prof_summary.py "$DB" "gpurun_out/${TAG}_tome_kernels.md" | grep -i "sttm\|kernel\|---"
cat "$REPO/gpurun_out/${TAG}_tome_probe.log"
