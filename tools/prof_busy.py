#!/usr/bin/env python3
"""GPU busy fraction of a rocprofv3 --kernel-trace run: union of all kernel intervals over [first start, last end] of the sttm kernels,
idle gaps by length, and how many kernels overlap on average.   python tools/prof_busy.py <results.db> [out.md]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = sorted(con.execute("select start, end, name from kernels where name like '%sttm%'"))
if not rows:
    print("no sttm kernels"); sys.exit(0)
# drop the warm-up half (first-touch allocations of the caching allocator stall the host for milliseconds): measure the steady part
rows = rows[len(rows) // 2:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = t1 - t0
ksum = sum(e - s for s, e, _ in rows)
big = [g for g in gaps if g > 20000]
text = (f"sttm kernels: {len(rows)}, span {tot / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms = {100 * busy / tot:.1f} %, "
        f"sum of durations / span = {ksum / tot:.2f} kernels in flight on average\n"
        f"idle gaps: {len(gaps)} totalling {sum(gaps) / 1e6:.3f} ms; gaps > 20 us: {len(big)} totalling {sum(big) / 1e6:.3f} ms "
        f"(largest {max(gaps) / 1e3 if gaps else 0:.1f} us)\n")
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
