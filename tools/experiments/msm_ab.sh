# same-box A/B of two builds of the library (LD_LIBRARY_PATH wins over the binary's RUNPATH)
python -m pytest tests/test_gpu_msm_table.py tests/test_gpu_parity.py -x -q -m gpu -k "msm" 2>&1 | tail -2
for round in 1 2; do
  for lib in new old; do
    echo "== $lib"
    if [ $lib = old ]; then export LD_LIBRARY_PATH=tools/experiments/oldlib; else unset LD_LIBRARY_PATH; fi
    ./tools/h2bench msmt 20 254 10 | grep msmt | cut -c1-200; ./tools/h2bench msmt 22 254 5 | grep msmt | cut -c1-200; ./tools/h2bench msmt 24 254 3 | grep msmt | cut -c1-200
  done
done
unset LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
H2BENCH_SPIN_MS=0 rocprofv3 --kernel-trace -d $R/gpurun_out/ptrace_new -o t -- $R/tools/h2bench msmt 24 254 2 > /dev/null 2>&1
cd $R
python3 tools/experiments/split_summary.py "$(find gpurun_out/ptrace_new -name '*results.db' | head -1)" | sed -n '/after k_table_build/,$p' | head -8
