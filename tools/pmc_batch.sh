#!/bin/bash
# GPU box: HBM-side traffic of the batch entry point's kernels PER LAUNCH SET (8 videos per launch, 3 internal streams), with and
# without the column-walk spatial stage: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only) over a short run of
# tools/colwalk_sweep.py.  FETCH_SIZE is in units of 64 bytes tallied per 128-byte request on gfx950: x2 (MI355X_MICROARCH.md, HBM).
# usage: tools/pmc_batch.sh <tag>      (environment: T, C, DT as in colwalk_sweep.py)
set -u
TAG=${1:-pmcb}; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
for CW in 0 1; do
  DBS=""
  for CNT in FETCH_SIZE WRITE_SIZE; do
    cd /tmp; rm -rf /tmp/pb_${CW}_$CNT
    N=192 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex sttm -d /tmp/pb_${CW}_$CNT -o x -- \
        python "$REPO/tools/colwalk_sweep.py" col_walk=$CW > /dev/null 2> "$REPO/gpurun_out/${TAG}_cw${CW}_$CNT.err"
    DBS="$DBS $(find /tmp/pb_${CW}_$CNT -name '*.db' | head -1)"
    cd "$REPO"
  done
  python tools/pmc_table.py "gpurun_out/${TAG}_cw${CW}_pmc.md" $DBS > /dev/null
  echo "== col_walk=$CW (bytes per launch; launches of the batch path cover 8 videos; FETCH_SIZE KB x 2 on gfx950)"; cat "gpurun_out/${TAG}_cw${CW}_pmc.md"
done
