#!/usr/bin/env python3
"""per-kernel averages before / after the first launch of a marker kernel (default k_table_build) in a rocprofv3 rocpd db.
usage: split_summary.py results.db [marker]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_table_build"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, duration, grid_x from kernels order by start"))
cut = next((r[1] for r in rows if marker in r[0]), None)
for label, sel in (("before", [r for r in rows if cut is None or r[1] < cut]), ("after", [r for r in rows if cut is not None and r[1] > cut])):
    agg = {}
    for name, _, dur, _ in sel:
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        a = agg.setdefault(name, [0, 0.0, 1e30])
        a[0] += 1
        a[1] += dur / 1e3
        a[2] = min(a[2], dur / 1e3)
    tot = sum(a[1] for a in agg.values()) or 1
    print("== %s %s: %.1f us of kernels" % (label, marker, tot))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-40s calls %5d  total %10.1f us  avg %9.2f us  min %9.2f us  %5.1f%%" % (name[:40], a[0], a[1], a[1] / a[0], a[2], 100 * a[1] / tot))
