#!/usr/bin/env python3
"""Kernel stats of a rocprofv3 run restricted to launches whose duration lies in [lo_us, hi_us] -- to read one problem size out of a
run that mixes several:  python tools/prof_range.py <results.db> <name substring> <lo_us> <hi_us>"""
import sqlite3, sys
db, sub, lo, hi = sys.argv[1], sys.argv[2], float(sys.argv[3]) * 1e3, float(sys.argv[4]) * 1e3
con = sqlite3.connect(db)
print("| kernel | launches | avg us | min us | max us |\n|---|---|---|---|---|")
for name, n, avg, mn, mx in con.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels "
                                        "where name like ? and duration between ? and ? group by name order by avg(duration)", (f"%{sub}%", lo, hi)):
    print(f"| {name.split('(')[0].replace('void ', '')} | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")
