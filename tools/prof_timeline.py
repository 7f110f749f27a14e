#!/usr/bin/env python3
"""Launch-by-launch view of a rocprofv3 --kernel-trace run (rocpd sqlite): per sttm kernel the duration percentiles, the mean
duration by position in the video pool (launch index mod POOL: content dependence shows as a period), and the gaps between the
end of one kernel and the start of the next on the timeline.     python tools/prof_timeline.py <results.db> [pool=8] [out.md]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    pool = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = list(con.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}"))
    rows = [(re.sub(r"\(.*", "", n).replace("void ", ""), s, e) for n, s, e in rows]
    out = []
    names = []
    for n, _, _ in rows:
        if "sttm::" in n and n not in names:
            names.append(n)
    out.append("| kernel | calls | p5 us | p50 us | p95 us | mean us | mean by launch index mod %d |" % pool)
    out.append("|---|---|---|---|---|---|---|")
    for n in names:
        d = [(e - s) / 1e3 for nn, s, e in rows if nn == n]
        if len(d) < 8:
            continue
        d = d[len(d) // 8:]                      # skip the warm-up part
        sd = sorted(d)
        by = [[] for _ in range(pool)]
        for i, v in enumerate(d):
            by[i % pool].append(v)
        out.append(f"| {n[:60]} | {len(d)} | {sd[len(sd) // 20]:.2f} | {sd[len(sd) // 2]:.2f} | {sd[len(sd) * 19 // 20]:.2f} | {sum(d) / len(d):.2f} | "
                   + " ".join(f"{sum(b) / max(1, len(b)):.1f}" for b in by) + " |")
    # gaps: end of kernel i -> start of kernel i + 1, grouped by (prev, next) name pair
    gaps = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        if "sttm::" in n0 and "sttm::" in n1:
            gaps.setdefault((n0[:28], n1[:28]), []).append((s1 - e0) / 1e3)
    out.append("")
    out.append("| end of -> start of | n | p50 gap us | mean gap us |")
    out.append("|---|---|---|---|")
    for k, g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
        sg = sorted(g)
        out.append(f"| {k[0]} -> {k[1]} | {len(g)} | {sg[len(sg) // 2]:.2f} | {sum(g) / len(g):.2f} |")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")


if __name__ == "__main__":
    main()
