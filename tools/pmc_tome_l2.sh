#!/bin/bash
# GPU box (via gpurun): L2 behaviour of the ToMe match kernel, first iteration at T = 128 (12544 x 12544 x 1024): L2 hits / misses and the
# read requests that leave the L2 towards the Infinity Cache / HBM (TCC_EA0_RDREQ: 64 B each, 32 B ones counted separately).
# Counters-only passes.  usage: tools/pmc_tome_l2.sh <tag> ["2 7" = tome_split modes] [extra env, e.g. STTM_TOME_ORDER=1]
set -u
TAG=${1:-l2}; MODES=${2:-"2"}; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
OUT="$REPO/gpurun_out/${TAG}_tome_l2.md"
echo "ToMe match kernel and the L2 (tools/pmc_tome_l2.sh): first iteration at T = 128; operands 2 x 25.7 MB (bf16) / 2 x 51.4 MB (fp32: two fp16 planes)" > "$OUT"; echo "" >> "$OUT"
echo "| input | tome_split | us | L2 requests | hit % | requests to the fabric (TCC_EA0_RDREQ) | of them 32 B | MB from beyond the L2 | MB the workgroups read |" >> "$OUT"
echo "|---|---|---|---|---|---|---|---|---|" >> "$OUT"
for DT in bfloat16 float32; do
  for SP in $MODES; do
    cd /tmp; rm -rf /tmp/l2_a /tmp/l2_b
    N_IT=6 DTYPE=$DT STTM_TOME_SPLIT=$SP timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex k_tome_match -d /tmp/l2_a -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
    N_IT=6 DTYPE=$DT STTM_TOME_SPLIT=$SP timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-include-regex k_tome_match -d /tmp/l2_b -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
    cd "$REPO"
    python - "$DT" "$SP" "$(find /tmp/l2_a -name '*.db' | head -1)" "$(find /tmp/l2_b -name '*.db' | head -1)" >> "$OUT" <<'PY'
import sqlite3, sys
dt, sp, da, db = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
def counters(path):
    vals = {}
    for n, v in sqlite3.connect(path).execute("select counter_name, value from counters_collection"):
        vals.setdefault(n, []).append(v)
    return lambda k: max(vals.get(k, [0.0]))           # the first iteration is the largest launch
a, b = counters(da), counters(db)
d = sorted(r[0] for r in sqlite3.connect(da).execute("select duration from kernels where name like '%k_tome_match%'"))
planes = 1 if dt != "float32" else 2
wg_mb = 49 * 49 * 2 * 256 * 1024 * 2 * planes / 1e6
rd, rd32 = b("TCC_EA0_RDREQ_sum"), b("TCC_EA0_RDREQ_32B_sum")
hit, miss = a("TCC_HIT_sum"), a("TCC_MISS_sum")
print(f"| {dt} | {sp} | {d[-1] / 1e3:.0f} | {a('TCC_REQ_sum'):.3g} | {100 * hit / max(hit + miss, 1):.1f} | {rd:.3g} | {rd32:.3g} | {((rd - rd32) * 64 + rd32 * 32) / 1e6:.0f} | {wg_mb:.0f} |")
PY
  done
done
cat "$OUT"
