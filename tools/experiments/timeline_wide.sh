cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tlw; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/pl -o m -- python3 tools/wide_bench.py ${1:-20} ${2:-16} > $out/wide.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/pl -name '*results.db' | head -1)" ${3:-2500} 0 > $out/timeline_wide.txt
python3 tools/rocprof_summary.py "$(find $out/pl -name '*results.db' | head -1)" > $out/wide_kernel_stats.txt 2>/dev/null
rm -rf $out/pl
