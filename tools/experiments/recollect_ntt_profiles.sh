#!/bin/bash
# the profiles that name k_ntt_pass, re-collected after a change to that kernel (a subset of tools/collect_profiles.sh)
set -u
out=gpurun_out/r4b_profiles; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
./tools/h2bench ntt 24 20 ntt 25 10 ntt 22 20 ntt 20 40 > "$out/h2bench_ntt.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/p_bench" -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/bench_line_profiled.json" 2>/dev/null
python3 tools/rocprof_summary.py "$(find "$out/p_bench" -name '*results.db' | head -1)" "$out/bench_kernel_stats.txt" > /dev/null
CMD="./tools/h2bench ntt 24 2 msm 20 254 2 eval 25 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/p_f" -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/p_w" -o w -- $CMD > /dev/null 2>&1
python3 tools/hbm_traffic.py "$(find "$out/p_f" -name '*results.db' | head -1)" "$(find "$out/p_w" -name '*results.db' | head -1)" "$CMD" > "$out/hbm_traffic.json"
rocprofv3 --kernel-trace --stats -d "$out/p_ntt" -o ntt -- python3 bench.py --steps 20 --warmup 5 --no-msm --prove-k 0 --k24 0 --wide-k 0 --no-cpu-baseline > "$out/bench_ntt_only_line.json" 2>/dev/null
python3 tools/rocprof_summary.py "$(find "$out/p_ntt" -name '*results.db' | head -1)" "$out/bench_ntt_only_kernel_stats.txt" > /dev/null
bash tools/experiments/nttpmc.sh > "$out/ntt_pass_pmc.txt" 2>&1
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 "$out"/p_*
