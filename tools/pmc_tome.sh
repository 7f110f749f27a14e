#!/bin/bash
# Run on the GPU box (via gpurun): MFMA / LDS counters of the ToMe match kernels (counters only, two passes),
# summary -> gpurun_out/<tag>_tome_pmc.md.   usage: tools/pmc_tome.sh <tag> [tome_split mode, default 1]
set -u
TAG=${1:-pmc}; MODE=${2:-1}
# DTYPE=bfloat16 tools/pmc_tome.sh <tag> 16   -> the one-plane (bf16 / fp16) kernel: one product term per score
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
rm -rf /tmp/pmc_tome1 /tmp/pmc_tome2
SPLIT=$MODE; [ "$MODE" = 16 ] && SPLIT=1
STTM_TOME_SPLIT=$SPLIT timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 \
    --kernel-include-regex k_tome_match -d /tmp/pmc_tome1 -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2> "$REPO/gpurun_out/pmc_tome.err"
STTM_TOME_SPLIT=$SPLIT timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS \
    --kernel-include-regex k_tome_match -d /tmp/pmc_tome2 -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2>> "$REPO/gpurun_out/pmc_tome.err"
cd "$REPO"
python - "$TAG" "$MODE" $(find /tmp/pmc_tome1 /tmp/pmc_tome2 -name '*.db') <<'PY'
import sqlite3, sys
tag, mode, dbs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
vals = {}
for db in dbs:
    con = sqlite3.connect(db)
    for n, c, a, m in con.execute("select counter_name, count(*), avg(value), max(value) from counters_collection group by counter_name"):
        vals[n] = (c, a, m)
import os
lines = [f"ToMe match kernel counters, tome_split = {mode} (tools/pmc_tome.sh; T=128, 14x14x1024 {os.environ.get('DTYPE', 'fp32')})", "",
         "| counter | launches | mean per launch | max (the 12544 x 12544 x 1024 first iteration) |", "|---|---|---|---|"]
for n, (c, a, m) in sorted(vals.items()):
    lines.append(f"| {n} | {c} | {a:.4g} | {m:.4g} |")
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs
    busy, gui = vals["SQ_VALU_MFMA_BUSY_CYCLES"][2], vals["GRBM_GUI_ACTIVE"][2] / 8.0
    terms = {0: 0, 1: 4, 2: 3, 3: 4, 4: 4, 5: 3, 6: 3, 16: 1}.get(mode, 4)
    if terms:
        n_mfma = terms * 2.0 * 12544 * 12544 * 1024 / 32768       # v_mfma_f32_32x32x16_f16: 32768 flop, 8 passes = 32 cycles
        lines.append("")
        lines.append(f"largest launch: {gui:.4g} active cycles per XCD; {terms} x 2*12544*12544*1024 / 32768 flop = {n_mfma:.3g} MFMA -> "
                     f"{busy / n_mfma:.1f} busy cycles per MFMA")
    else:
        n_mfma = 2.0 * 12544 * 12544 * 1024 / 4096
        lines.append("")
        lines.append(f"largest launch: {gui:.4g} active cycles per XCD; {n_mfma:.3g} fp32-input MFMA -> {busy / n_mfma:.1f} busy cycles per MFMA")
    lines.append(f"MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 256 CUs x 4 SIMDs) = {100.0 * busy / (gui * 1024):.1f} %")
    if "SQ_LDS_BANK_CONFLICT" in vals and "SQ_LDS_IDX_ACTIVE" in vals:
        lines.append(f"LDS: bank-conflict cycles / active cycles = {100.0 * vals['SQ_LDS_BANK_CONFLICT'][2] / max(1.0, vals['SQ_LDS_IDX_ACTIVE'][2]):.1f} %; "
                     f"LDS active / (cycles x 256 CUs) = {100.0 * vals['SQ_LDS_IDX_ACTIVE'][2] / (gui * 256):.1f} %")
text = "\n".join(lines)
open(f"gpurun_out/{tag}_tome_pmc.md", "w").write(text + "\n")
print(text)
PY
