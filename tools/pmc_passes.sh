#!/bin/bash
# Run on the GPU box (via gpurun): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, sttm kernels only)
# over a short bench.py run, then tools/pmc_summary.py -> profiles/pmc_traffic.json + gpurun_out/<tag>_pmc_traffic.md
# usage: tools/pmc_passes.sh <tag>
set -u
TAG=${1:-pmc}
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
for CNT in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    rm -rf /tmp/pmc_$CNT
    timeout 600 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex sttm -d /tmp/pmc_$CNT -o x -- \
        python "$REPO/bench.py" --mode dropin --steps 2 --warmup 1 --videos-per-step 64 --profile-calls 8 --no-cpu-baseline --no-extensions --no-configs > /dev/null 2> "$REPO/gpurun_out/pmc_$CNT.err"
    cp "$(find /tmp/pmc_$CNT -name '*.db' | head -1)" "$REPO/gpurun_out/pmc_$CNT.db"
    cd "$REPO"
done
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE.db gpurun_out/pmc_WRITE_SIZE.db "$TAG" | tee "gpurun_out/${TAG}_pmc_traffic.md"
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
