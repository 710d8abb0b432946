cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/msmprof; mkdir -p $out
for L in "$@"; do
  rocprofv3 --kernel-trace -d $out/p_$L -o m -- ./tools/h2bench msmt $L 254 3 > $out/h2bench_t$L.txt 2>/dev/null
  python3 tools/experiments/split_summary.py "$(find $out/p_$L -name '*results.db' | head -1)" > $out/msmt_${L}_split.txt
done
rm -rf $out/p_*
