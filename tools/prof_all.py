#!/usr/bin/env python3
"""All kernels of a rocprofv3 rocpd database, by total time (no filtering): python tools/prof_all.py <results.db> [n]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                        "group by name order by sum(duration) desc"))
for name, calls, tot, avg, mn, mx in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print(f"{name[:110]:110s} {calls:6d} {tot / 1e6:9.3f} ms  avg {avg / 1e3:8.2f}  min {mn / 1e3:8.2f}  max {mx / 1e3:8.2f} us")
