#!/bin/bash
# GPU box, DEVELOPMENT build (python -m sttm_amd.build --dev): the running max of the 256-tile ToMe match kernels, one-pass sweep (STTM_TOME_ABL = 8:
# round, compare, two selects per score) against the two-pass form (0: fp32 maximum, one rounding per row, first candidate at or above the tie
# floor only when a lane improves), interleaved on ONE box.  Kernel time from rocprofv3 --kernel-trace, first iteration at T = 128; outputs of both
# forms are bit-identical (tests/test_hip_parity.py runs every ToMe case on the product build; tools/tome_ab_w4.py compares whole calls).
# usage: tools/tome_epilogue_ab.sh <tag>
TAG=${1:-epi}; REPO=$(pwd); export TMPDIR=/tmp STTM_LIB=dev; OUT="$REPO/gpurun_out/${TAG}_tome_epilogue_ab.md"
echo "| input | tome_split (2 eight waves, 7 four) | one-pass sweep, us | two passes, us | (repeat) sweep | two passes |" > "$OUT"; echo "|---|---|---|---|---|---|" >> "$OUT"
for DT in bfloat16 float16 float32; do
  for SP in 2 7; do
    ROW="| $DT | $SP |"
    for ABL in 8 0 8 0; do
      cd /tmp; rm -rf /tmp/ab_t
      N_IT=6 DTYPE=$DT STTM_TOME_SPLIT=$SP STTM_TOME_ABL=$ABL timeout 300 rocprofv3 --kernel-trace --kernel-include-regex k_tome_match -d /tmp/ab_t -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
      cd "$REPO"
      ROW="$ROW $(python - "$(find /tmp/ab_t -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
d = sorted(r[0] for r in sqlite3.connect(sys.argv[1]).execute("select duration from kernels where name like '%k_tome_match%'"))
top = d[-max(1, len(d) // 4):]
print(f"{top[len(top) // 2] / 1e3:.0f}")
PY
) |"
    done
    echo "$ROW" >> "$OUT"
  done
done
cat "$OUT"
