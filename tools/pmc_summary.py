#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, sttm kernels only) into
profiles/pmc_traffic.json (read by bench.py for roofline.traffic) and a markdown table.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE.db gpurun_out/pmc_WRITE_SIZE.db <tag>

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as reported (KB)."""
import json
import os
import re
import sqlite3
import sys

NAMES = {"k_spatial": "quadtree_spatial", "k_pairs": "temporal_pairs_labels", "k_pairs256": "temporal_pairs_labels", "k_col_labels": "labels_standalone",
         "k_group_mean": "group_mean", "k_slow_filter": "temporal_pairs_labels"}


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    out = {}
    for name, n, avg in con.execute("select kernel_name, count(*), avg(value) from counters_collection "
                                    "where counter_name = ? group by kernel_name", (counter,)):
        m = re.search(r"sttm::(k_[a-z_0-9]+)", name)
        if m:
            out.setdefault(m.group(1), []).append((name, n, avg))
    return out


def main():
    fetch_db, write_db, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    rows, agg = [], {}
    for k in sorted(set(f) | set(w)):
        fk = sum(a for _, _, a in f.get(k, []))
        wk = sum(a for _, _, a in w.get(k, []))
        fetch_b, write_b = 2 * fk * 1024, wk * 1024
        rows.append((k, f.get(k, [("", 0, 0)])[0][1], fk, fetch_b / 1e6, write_b / 1e6))
        g = NAMES.get(k, k)
        agg[g] = agg.get(g, 0) + fetch_b + write_b
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sttm_amd import _lib
    rec = {"workload": "T128_14x14x1024_f32_sttm_0.85_0.55", "tag": tag, "build_tag": _lib.build_tag(),
           "hbm_bytes_per_video": round(sum(agg.values())),
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-include-regex sttm, "
                     "bench.py --mode dropin --steps 2 --warmup 1 --videos-per-step 64 (one video per launch); mean per launch; fetch = 2 * FETCH_SIZE KB (gfx950 correction), write = WRITE_SIZE KB",
           "hbm_bytes_per_launch": {k: round(v) for k, v in agg.items()}}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    lines = ["| kernel | launches | FETCH_SIZE KB (raw) | fetch MB (x2 corrected) | write MB |", "|---|---|---|---|---|"]
    for k, n, fk, fm, wm in rows:
        lines.append(f"| {k} | {n} | {fk:.1f} | {fm:.1f} | {wm:.1f} |")
    lines.append("")
    lines.append("total HBM bytes per video: %.1f MB (algorithmic B = 148.9 MB)" % (sum(agg.values()) / 1e6))
    text = "\n".join(lines)
    with open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.md"), "w") as fh:
        fh.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
