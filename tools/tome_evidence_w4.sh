#!/bin/bash
# GPU box (via gpurun): the eight-wave (tome_split = 6) and the four-wave (7) form of the 256-tile ToMe match kernel on ONE box, first
# iteration at T = 128 (12544 x 12544 x 1024): kernel time (rocprofv3 --kernel-trace), MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE,
# a counters-only pass) and where the waves' cycles go (SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY, a third pass).
# usage: tools/tome_evidence_w4.sh <tag> ["6 7 2" = tome_split modes]     -> gpurun_out/<tag>_tome_w4.md
set -u
TAG=${1:-ev}; MODES=${2:-"6 7"}; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
OUT="$REPO/gpurun_out/${TAG}_tome_w4.md"
echo "ToMe match kernel, first iteration at T=128 (12544 x 12544 x 1024), one box (tools/tome_evidence_w4.sh); fp32 = three fp16 plane products per score" > "$OUT"
echo "" >> "$OUT"
echo "| input | tome_split | kernel | us (trace pass) | clock GHz | busy cycles / MFMA | MfmaUtil % | wave cycles: waiting % | issue-stalled % | issuing % |" >> "$OUT"
echo "|---|---|---|---|---|---|---|---|---|---|" >> "$OUT"
for DT in bfloat16 float32; do
  for SP in $MODES; do
    cd /tmp; rm -rf /tmp/ev_t /tmp/ev_c /tmp/ev_w
    N_IT=6 DTYPE=$DT STTM_TOME_SPLIT=$SP timeout 300 rocprofv3 --kernel-trace --kernel-include-regex k_tome_match -d /tmp/ev_t -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
    N_IT=6 DTYPE=$DT STTM_TOME_SPLIT=$SP timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex k_tome_match -d /tmp/ev_c -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
    N_IT=6 DTYPE=$DT STTM_TOME_SPLIT=$SP timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex k_tome_match -d /tmp/ev_w -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
    cd "$REPO"
    python - "$DT" "$SP" "$(find /tmp/ev_t -name '*.db' | head -1)" "$(find /tmp/ev_c -name '*.db' | head -1)" "$(find /tmp/ev_w -name '*.db' | head -1)" >> "$OUT" <<'PY'
import re, sqlite3, sys
dt, sp, dbt, dbc, dbw = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
con = sqlite3.connect(dbt)
rows = sorted((r[0], r[1]) for r in con.execute("select duration, name from kernels where name like '%k_tome_match%'"))
top = rows[-max(1, len(rows) // 4):]
us = top[len(top) // 2][0] / 1e3
name = re.sub(r"\(.*$", "", top[-1][1]).replace("sttm::", "").replace("void ", "")
def counters(db):
    vals = {}
    for n, v in sqlite3.connect(db).execute("select counter_name, value from counters_collection"):
        vals.setdefault(n, []).append(v)
    def top_med(k):
        v = sorted(vals.get(k, [0.0])); t = v[-max(1, len(v) // 4):]; return t[len(t) // 2]
    return top_med
c, w = counters(dbc), counters(dbw)
gui, busy = c("GRBM_GUI_ACTIVE") / 8.0, c("SQ_VALU_MFMA_BUSY_CYCLES")
terms = 1 if dt != "float32" else (3 if sp in (2, 5, 6, 7) else 4)
n_mfma = terms * 2.0 * 12544 * 12544 * 1024 / 32768
wc = max(w("SQ_WAVE_CYCLES"), 1.0)
print(f"| {dt} | {sp} | {name} | {us:.0f} | {gui / (us * 1e3):.2f} | {busy / n_mfma:.1f} | {100.0 * busy / (gui * 1024):.1f} | "
      f"{100 * w('SQ_WAIT_ANY') / wc:.0f} | {100 * w('SQ_WAIT_INST_ANY') / wc:.0f} | {100 * w('SQ_ACTIVE_INST_ANY') / wc:.0f} |")
PY
  done
done
echo "" >> "$OUT"
echo "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); a v_mfma_f32_32x32x16 is 8 passes = 32 cycles at full rate; waiting = SQ_WAIT_ANY (s_waitcnt / barrier), issue-stalled = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY, all / SQ_WAVE_CYCLES." >> "$OUT"
cat "$OUT"
