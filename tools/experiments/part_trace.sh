cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export H2BENCH_SPIN_MS=0
run() { # name, mask
  if [ -n "$2" ]; then export H2BENCH_CUMASK=$2; else unset H2BENCH_CUMASK; fi
  rocprofv3 --kernel-trace -d $R/gpurun_out/ptrace_$1 -o t -- $R/tools/h2bench msmt 22 254 1 > $R/gpurun_out/ptrace_$1.log 2>&1
  echo "== $1"; grep msmt $R/gpurun_out/ptrace_$1.log | cut -c1-120
  ( cd $R; python3 tools/experiments/split_summary.py "$(find gpurun_out/ptrace_$1 -name '*results.db' | head -1)" | sed -n '/after k_table_build/,$p' | head -7 )
}
run full ""
run contig ffffffff,0,0,0,0,0,0,0
run inter 01010101,01010101,01010101,01010101,01010101,01010101,01010101,01010101
run half ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0
