cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/adv; mkdir -p $out
export H2BENCH_ADVICE=1
rocprofv3 --kernel-trace --stats -d $out/p -o m -- ./tools/h2bench msm 24 16 3 > $out/h2bench.txt 2>/dev/null
python3 tools/rocprof_summary.py "$(find $out/p -name '*results.db' | head -1)" $out/stats.txt > /dev/null
rm -rf $out/p
