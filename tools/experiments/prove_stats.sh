cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ps; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/p -o m -- python3 tools/prove_bench.py ${1:-22} 3 > $out/prove.txt 2>/dev/null
python3 tools/rocprof_summary.py "$(find $out/p -name '*results.db' | head -1)" $out/stats_${1:-22}.txt > /dev/null
rm -rf $out/p
