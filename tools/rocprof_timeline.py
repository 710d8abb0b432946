#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 rocpd database: the last `count` dispatches before the end (or from `skip`), with
start offsets, durations and the idle gap before each -- shows where a latency-bound sequence spends its time.
usage: rocprof_timeline.py results.db [count] [skip_from_end]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    rows = rows[len(rows) - skip - count:len(rows) - skip] if skip + count <= len(rows) else rows
    t0, prev_end = rows[0][1], rows[0][1]
    for name, start, end, grid, wg in rows:
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print("%9.1f us  +%7.1f gap  %8.1f us  %-40s grid %d / %d" % ((start - t0) / 1e3, (start - prev_end) / 1e3, (end - start) / 1e3,
                                                                 name[:40], grid, wg))
        prev_end = max(prev_end, end)


if __name__ == "__main__":
    main()
