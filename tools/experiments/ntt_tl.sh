cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ntl; mkdir -p $out
for c in "$@"; do
  H2_NTT_LOGC=$c rocprofv3 --kernel-trace -d $out/p$c -o m -- ./tools/h2bench ntt 24 6 > $out/h2bench_$c.txt 2>/dev/null
  python3 tools/rocprof_timeline.py "$(find $out/p$c -name '*results.db' | head -1)" 12 0 > $out/timeline_$c.txt
done
rm -rf $out/p*
