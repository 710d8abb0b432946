cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/proveprof; mkdir -p $out
K=$1
for T in 1 0; do
  export H2_MSM_TABLES=$T
  rocprofv3 --kernel-trace --stats -d $out/p_$T -o m -- python3 tools/prove_bench.py $K 3 > $out/prove_${K}_tables$T.txt 2>/dev/null
  python3 tools/rocprof_summary.py "$(find $out/p_$T -name '*results.db' | head -1)" $out/prove_${K}_tables${T}_kernels.txt > /dev/null
done
rm -rf $out/p_*
