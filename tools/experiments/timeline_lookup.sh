#!/bin/bash
# kernel timeline + per-kernel stats of the k = 18 lookup + shuffle proof (tools/lookup_bench.py): where a mid-size proof's
# device time goes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tll; mkdir -p $out
python3 tools/lookup_bench.py ${1:-18} > $out/lookup_plain.txt 2>/dev/null
rocprofv3 --kernel-trace --stats -d $out/pl -o m -- python3 tools/lookup_bench.py ${1:-18} > $out/lookup.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/pl -name '*results.db' | head -1)" ${2:-600} 0 > $out/timeline_lookup.txt
python3 tools/rocprof_summary.py "$(find $out/pl -name '*results.db' | head -1)" > $out/lookup_kernel_stats.txt 2>/dev/null
rm -rf $out/pl
