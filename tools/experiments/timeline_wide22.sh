cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tlw22; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/pl -o m -- python3 tools/wide_bench.py 22 16 - compact > $out/wide.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/pl -name '*results.db' | head -1)" 60000 0 > $out/timeline_wide.txt
python3 tools/rocprof_summary.py "$(find $out/pl -name '*results.db' | head -1)" > $out/wide_kernel_stats.txt 2>/dev/null
rm -rf $out/pl
