#!/usr/bin/env python3
"""Kernel stats of a rocprofv3 run grouped by (kernel, grid size) for launches within [lo_us, hi_us] -- tells apart two work splits of
the same kernel:  python tools/prof_by_grid.py <results.db> <name substring> <lo_us> <hi_us>"""
import sqlite3, sys
db, sub, lo, hi = sys.argv[1], sys.argv[2], float(sys.argv[3]) * 1e3, float(sys.argv[4]) * 1e3
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
grid = next((c for c in ("grid_x", "grid_size_x", "grid_size", "workgroup_count_x") if c in cols), None)
if grid is None:
    print("columns:", cols); sys.exit(1)
print("| kernel | grid | launches | avg us | min us | max us |\n|---|---|---|---|---|---|")
for name, g, n, avg, mn, mx in con.execute(f"select name, {grid}, count(*), avg(duration), min(duration), max(duration) from kernels "
                                           f"where name like ? and duration between ? and ? group by name, {grid} order by name, avg(duration)",
                                           (f"%{sub}%", lo, hi)):
    print(f"| {name.split('(')[0].replace('void ', '')} | {g} | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")
