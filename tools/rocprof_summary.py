#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`) into
the per-kernel summary text kept under profiles/.   usage: rocprof_summary.py results.db [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)  # drop the argument list
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 70 else name[:67] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                            "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
                            "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    lines = ["%-70s %6s %12s %11s %11s %11s %6s %5s %5s %7s %5s %9s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "wg", "grid_x")]
    for r in rows:
        lines.append("%-70s %6d %12.1f %11.2f %11.2f %11.2f %6.2f %5d %5d %7d %5d %9d" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, (r[6] or 0) + (r[7] or 0), r[8] or 0,
            r[9] or 0, r[10] or 0, r[11] or 0))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
