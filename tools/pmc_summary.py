#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass:  python tools/pmc_summary.py <results.db> [substr]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name")
out = {}
for name, counter, total, launches in rows:
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if flt in name:
        out.setdefault(name, {})[counter] = total / launches
        out[name]["launches"] = launches
for name in sorted(out):
    print(name)
    for k in sorted(out[name]):
        print("    %-28s %16.1f" % (k, out[name][k]))
