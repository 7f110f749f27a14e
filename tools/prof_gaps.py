#!/usr/bin/env python3
"""Idle time between the consecutive STTM kernels of a rocprofv3 --kernel-trace run (rocpd sqlite): for every adjacent pair
(previous kernel -> next kernel) the mean gap between the end of one and the start of the next.

    python tools/prof_gaps.py <results.db>
"""
import re
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
s_col = "start" if "start" in cols else [c for c in cols if "start" in c][0]
e_col = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = list(con.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}"))
gaps = defaultdict(list)
short = lambda n: re.sub(r"<.*", "", n.replace("void ", "").replace("sttm::", ""))
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    if "sttm::" in n0 and "sttm::" in n1:
        gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
print("| from -> to | pairs | mean gap us | median | min |")
print("|---|---|---|---|---|")
for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    g.sort()
    print(f"| {a} -> {b} | {len(g)} | {sum(g) / len(g):.2f} | {g[len(g) // 2]:.2f} | {g[0]:.2f} |")
