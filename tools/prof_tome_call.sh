#!/bin/bash
# GPU box (via gpurun): per-kernel times of whole get_tome_features calls (ratio 0.5 / 0.7 / 0.85) at T frames, rocprofv3 --kernel-trace.
# usage: tools/prof_tome_call.sh <tag> [T=180] [DTYPE=bfloat16]
set -u
TAG=${1:-tc}; export T=${2:-180}; export DTYPE=${3:-bfloat16}; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
cd /tmp; rm -rf /tmp/tc_t
N_IT=8 timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace -d /tmp/tc_t -o x -- python "$REPO/tools/bench_tome.py" > /tmp/tc.log 2>&1
cd "$REPO"
python - "$(find /tmp/tc_t -name '*.db' | head -1)" > "gpurun_out/${TAG}_tome_call_T${T}_${DTYPE}.md" <<'PY'
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end from kernels order by start"))
# the first get_tome_features calls of the run are ratio 0.5: ONE iteration each; take the kernels between consecutive k_tome_normalize launches
calls, cur = [], None
for n, s, e in rows:
    short = re.sub(r"\(.*$", "", n).replace("sttm::", "").replace("void ", "")
    if "k_tome_normalize" in short:
        if cur: calls.append(cur)
        cur = []
    if cur is not None: cur.append((short, s, e))
if cur: calls.append(cur)
one = [c for c in calls if sum(1 for k in c if "k_tome_normalize" in k[0]) == 1 and any("k_tome_merge" in k[0] for k in c)][1:9]   # skip the warm-up call
print("| kernel | us (median over %d one-iteration calls) | gap before it, us |" % len(one)); print("|---|---|---|")
names = [k[0] for k in one[0] if "k_tome" in k[0]]
tot = 0.0
for i, nm in enumerate(names):
    d, g = [], []
    for c in one:
        ks = [k for k in c if "k_tome" in k[0]]
        if i < len(ks):
            d.append((ks[i][2] - ks[i][1]) / 1e3)
            if i: g.append((ks[i][1] - ks[i - 1][2]) / 1e3)
    d.sort(); g.sort()
    tot += d[len(d) // 2]
    print(f"| {nm[:70]} | {d[len(d) // 2]:.1f} | {g[len(g) // 2]:.1f} |" if g else f"| {nm[:70]} | {d[len(d) // 2]:.1f} | |")
span = sorted((max(k[2] for k in c if 'k_tome' in k[0]) - min(k[1] for k in c if 'k_tome' in k[0])) / 1e3 for c in one)
print(f"\nsum of kernel durations {tot:.1f} us; first start to last end of a call {span[len(span) // 2]:.1f} us")
PY
cat "gpurun_out/${TAG}_tome_call_T${T}_${DTYPE}.md"; tail -3 /tmp/tc.log
