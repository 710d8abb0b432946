cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tl; mkdir -p $out
rocprofv3 --kernel-trace -d $out/p -o m -- python3 tools/prove_bench.py ${1:-24} 2 > $out/prove.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/p -name '*results.db' | head -1)" 400 0 > $out/timeline_${1:-24}.txt
rm -rf $out/p
