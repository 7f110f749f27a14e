#!/bin/bash
# Run on the GPU box (via gpurun): MFMA counters of the ToMe match kernel (counters only, one pass), summary -> gpurun_out/<tag>_tome_pmc.md
set -u
TAG=${1:-pmc}
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
rm -rf /tmp/pmc_tome
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-include-regex k_tome_match \
    -d /tmp/pmc_tome -o x -- python "$REPO/tools/bench_tome.py" > /dev/null 2> "$REPO/gpurun_out/pmc_tome.err"
cd "$REPO"
python - "$(find /tmp/pmc_tome -name '*.db' | head -1)" "$TAG" <<'PY'
import sqlite3, sys
db, tag = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select counter_name, count(*), avg(value), max(value) from counters_collection group by counter_name"))
vals = {n: (c, a, m) for n, c, a, m in rows}
lines = ["| counter | launches | mean per launch | max (the 12544 x 12544 x 1024 first iteration) |", "|---|---|---|---|"]
for n, (c, a, m) in sorted(vals.items()):
    lines.append(f"| {n} | {c} | {a:.4g} | {m:.4g} |")
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs (its value / 8 / kernel duration gives the shader clock);
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs (64 cycles per 32x32x2 f32 MFMA: 16 passes x 4)
    busy, gui = vals["SQ_VALU_MFMA_BUSY_CYCLES"][2], vals["GRBM_GUI_ACTIVE"][2] / 8.0
    lines.append("")
    lines.append(f"largest launch: {gui:.4g} active cycles per XCD; MFMA instructions = 2*12544*12544*1024 / 4096 flop = 7.87e7 -> "
                 f"{busy / 7.87e7:.1f} busy cycles per MFMA")
    lines.append(f"MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 256 CUs x 4 SIMDs) = {100.0 * busy / (gui * 1024):.1f} %")
text = "\n".join(lines)
open(f"gpurun_out/{tag}_tome_pmc.md", "w").write(text + "\n")
print(text)
PY
