#!/bin/bash
# GPU box, DEVELOPMENT build (python -m sttm_amd.build --dev): the four-wave ToMe match kernel (tome_split = 7, bf16) and its ablations on one
# box: STTM_TOME_ABL = 0 the kernel, 7 no running max, 1 no DMA after the prologue, 6 MFMAs + fragment reads (no DMA, no barriers), 5 MFMAs alone
# (outputs invalid except for 0).  Kernel time from rocprofv3 --kernel-trace, first iteration at T = 128.
# usage: tools/tome_ablate_w4.sh <tag>
TAG=${1:-abl}; REPO=$(pwd); export TMPDIR=/tmp STTM_LIB=dev; OUT="$REPO/gpurun_out/${TAG}_tome_w4_ablation.md"
echo "| STTM_TOME_ABL | four-wave bf16 match kernel, us (T = 128 first iteration) |" > "$OUT"; echo "|---|---|" >> "$OUT"
for ABL in 0 7 1 6 5; do
  cd /tmp; rm -rf /tmp/ab_t
  N_IT=6 DTYPE=bfloat16 STTM_TOME_SPLIT=7 STTM_TOME_ABL=$ABL timeout 300 rocprofv3 --kernel-trace --kernel-include-regex k_tome_match -d /tmp/ab_t -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>&1
  cd "$REPO"
  python - "$ABL" "$(find /tmp/ab_t -name '*.db' | head -1)" >> "$OUT" <<'PY'
import sqlite3, sys
d = sorted(r[0] for r in sqlite3.connect(sys.argv[2]).execute("select duration from kernels where name like '%k_tome_match%'"))
top = d[-max(1, len(d) // 4):]
print(f"| {sys.argv[1]} | {top[len(top) // 2] / 1e3:.0f} |")
PY
done
cat "$OUT"
