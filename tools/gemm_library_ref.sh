#!/bin/bash
# GPU box (via gpurun): the vendor library's GEMM at the ToMe match shape beside the fused match kernel, ONE box: kernel name and resources
# (rocprofv3 --kernel-trace, csv), MfmaUtil and the wave-state split (counter passes).  usage: tools/gemm_library_ref.sh <tag>
set -u
TAG=${1:-lib}; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
OUT="$REPO/gpurun_out/${TAG}_gemm_library_ref.md"
echo "Vendor-library GEMM at the ToMe match shape (T = 128: 12544 x 12544 x 1024), one box (tools/gemm_library_ref.sh)" > "$OUT"; echo "" >> "$OUT"
for DT in bfloat16 float16; do
  cd /tmp; rm -rf /tmp/gl_t /tmp/gl_c /tmp/gl_w
  DTYPE=$DT TS=128,180 python "$REPO/tools/gemm_library_ref.py" >> "$OUT" 2>&1
  DTYPE=$DT TS=128 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gl_t -o x -- python "$REPO/tools/gemm_library_ref.py" > /dev/null 2>&1
  DTYPE=$DT TS=128 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/gl_c -o x -- python "$REPO/tools/gemm_library_ref.py" > /dev/null 2>&1
  DTYPE=$DT TS=128 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/gl_w -o x -- python "$REPO/tools/gemm_library_ref.py" > /dev/null 2>&1
  cd "$REPO"
  python - "$DT" /tmp/gl_t /tmp/gl_c /tmp/gl_w >> "$OUT" <<'PY'
import csv, glob, sys
dt, dt_, dc, dw = sys.argv[1:5]
def rows(d, pat):
    out = []
    for f in glob.glob(d + "/**/*" + pat + "*.csv", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
tr = rows(dt_, "kernel_trace")
dur = {}
for r in tr:
    dur.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
name = max(dur, key=lambda k: sorted(d for d, _ in dur[k])[len(dur[k]) // 2])      # the GEMM: the longest kernel by median
ds = sorted(d for d, _ in dur[name]); us = ds[len(ds) // 2] / 1e3
r0 = dur[name][0][1]
print(f"\n{dt}: dominant kernel `{name}`")
print("  resources: " + ", ".join(f"{k}={r0[k]}" for k in r0 if any(s in k for s in ("Workgroup_Size", "Grid_Size", "LDS", "Scratch", "VGPR", "SGPR"))))
def cnt(d):
    v = {}
    for r in rows(d, "counter_collection"):
        if r.get("Kernel_Name") == name:
            v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return lambda k: sorted(v.get(k, [0.0]))[len(v.get(k, [0.0])) // 2]
c, w = cnt(dc), cnt(dw)
gui, busy = c("GRBM_GUI_ACTIVE") / 8.0, c("SQ_VALU_MFMA_BUSY_CYCLES")
wc = max(w("SQ_WAVE_CYCLES"), 1.0)
print(f"  {us:.0f} us (trace pass, median of {len(ds)}), clock {gui / (us * 1e3):.2f} GHz, MfmaUtil {100.0 * busy / max(gui * 1024, 1):.1f} %, "
      f"waves waiting / issue-stalled / issuing {100 * w('SQ_WAIT_ANY') / wc:.0f} / {100 * w('SQ_WAIT_INST_ANY') / wc:.0f} / {100 * w('SQ_ACTIVE_INST_ANY') / wc:.0f} %")
PY
done
echo "" >> "$OUT"
bash "$REPO/tools/tome_evidence_w4.sh" "$TAG" "2" > /dev/null 2>&1
cat "$REPO/gpurun_out/${TAG}_tome_w4.md" >> "$OUT"
cat "$OUT"
