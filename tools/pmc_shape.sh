#!/bin/bash
# Run on the GPU box (via gpurun): FETCH_SIZE and WRITE_SIZE per sttm kernel (two rocprofv3 --pmc passes, counters only) for one
# tools/ab_variants.py shape, e.g.   bash tools/pmc_shape.sh c4 default
set -u
SHAPE=${1:-c4}; SPEC=${2:-default}
REPO=$(pwd); export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; rm -rf /tmp/pmcs_$CNT
  timeout 600 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex sttm -d /tmp/pmcs_$CNT -o x -- python "$REPO/tools/ab_variants.py" --shape $SHAPE $SPEC > /dev/null 2>&1
  DB=$(find /tmp/pmcs_$CNT -name '*.db' | head -1)
  echo "## $SHAPE $SPEC $CNT (KB per launch; fetch bytes = 2 x FETCH_SIZE KB on gfx950, write bytes = WRITE_SIZE KB)"
  python - "$DB" $CNT <<'PY'
import sqlite3,sys,re
con=sqlite3.connect(sys.argv[1]); tot=0
for name,n,avg in con.execute("select kernel_name,count(*),avg(value) from counters_collection where counter_name=? group by kernel_name",(sys.argv[2],)):
    m=re.search(r"sttm::(k_[a-z_0-9]+)",name)
    if m and n > 100:
        mb=(2 if sys.argv[2]=="FETCH_SIZE" else 1)*avg*1024/1e6; tot+=mb
        print(f"{m.group(1):20s} n={n:6d} avg {avg:12.1f} KB -> {mb:8.1f} MB")
print(f"total {tot:.1f} MB per video")
PY
  cd "$REPO"
done
