cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tl; mkdir -p $out
rocprofv3 --kernel-trace -d $out/pl -o m -- python3 tools/lookup_bench.py 18 > $out/lookup.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/pl -name '*results.db' | head -1)" 900 0 > $out/timeline_lookup18.txt
rm -rf $out/pl
