#!/usr/bin/env python3
"""Per-kernel time by videos per launch set from a rocprofv3 rocpd database of tools/bench_batch.py:
   python tools/prof_batch.py <results.db>   (grid_y / workgroup_y = videos in the launch)"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, grid_y / workgroup_y, count(*), avg(duration) from kernels where name like '%sttm%' "
                        "group by name, grid_y / workgroup_y order by name, grid_y / workgroup_y"))
for n, gy, c, a in rows:
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    print(f"{n[:56]:56s} videos/launch {gy:3d} calls {c:6d} avg {a / 1e3:8.2f} us  per video {a / 1e3 / max(gy, 1):7.2f} us")
