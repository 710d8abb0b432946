#!/usr/bin/env python3
"""HBM bytes per launch per kernel from two rocprofv3 PMC passes (one counter per pass, as gpurun requires):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o f -- <cmd>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o w -- <cmd>
    python tools/hbm_traffic.py A/f_results.db B/w_results.db "<cmd>" > profiles/<name>.json

Correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB; on gfx950 FETCH_SIZE counts a
128-byte request as 64 bytes, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- calibrated for wide coalesced reads;
for gather-heavy kernels (the MSM accumulation) it is an upper bound."""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, total, launches in db.execute(
            "select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name",
            (counter,)):
        out[re.sub(r"\(.*", "", name).replace("void ", "")] = (total / launches, launches)
    return out


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": max(nf, nw),
                      "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024}
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- " +
               (sys.argv[3] if len(sys.argv) > 3 else "<cmd>"),
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B; "
               "MI355X_MICROARCH.md section HBM); the x2 is calibrated for wide coalesced reads only, so gather-heavy MSM "
               "kernels are upper bounds",
               "kernels": kernels}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
