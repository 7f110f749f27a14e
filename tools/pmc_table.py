#!/usr/bin/env python3
"""Per-kernel averages of every counter found in a set of rocprofv3 --pmc result databases (one pass per database).
    python tools/pmc_table.py out.md pass1.db pass2.db ...   [--match REGEX]
Kernel names are shortened to the part that identifies them (template arguments kept)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name.replace("sttm::", "")[:90]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    for i, a in enumerate(sys.argv):
        if a == "--match":
            match = re.compile(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
    out, dbs = args[0], args[1:]
    table, counters = {}, []
    for db in dbs:
        con = sqlite3.connect(db)
        for name, cn, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                            "group by kernel_name, counter_name"):
            if match and not match.search(name):
                continue
            k = short(name)
            table.setdefault(k, {})[cn] = (n, avg)
            if cn not in counters:
                counters.append(cn)
    lines = ["| kernel | launches | " + " | ".join(counters) + " |", "|---|---|" + "---|" * len(counters)]
    for k in sorted(table):
        n = max(v[0] for v in table[k].values())
        lines.append(f"| {k} | {n} | " + " | ".join(("%.4g" % table[k][c][1]) if c in table[k] else "" for c in counters) + " |")
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
