set -u
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for SPEC in k1_split=-1 default; do
  for CNT in FETCH_SIZE; do
    cd /tmp; rm -rf /tmp/pmcc_$CNT
    timeout 600 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex sttm -d /tmp/pmcc_$CNT -o x -- python "$REPO/tools/ab_variants.py" --shape c4b $SPEC > /dev/null 2>&1
    DB=$(find /tmp/pmcc_$CNT -name '*.db' | head -1)
    echo "## $SPEC $CNT"
    python - "$DB" $CNT <<'PY'
import sqlite3,sys,re
con=sqlite3.connect(sys.argv[1])
for name,n,avg in con.execute("select kernel_name,count(*),avg(value) from counters_collection where counter_name=? group by kernel_name",(sys.argv[2],)):
    m=re.search(r"sttm::(k_[a-z_0-9]+)",name)
    if m: print(f"{m.group(1):20s} n={n:6d} avg {avg:12.1f} KB raw -> {2*avg*1024/1e6:8.1f} MB")
PY
    cd "$REPO"
  done
done
