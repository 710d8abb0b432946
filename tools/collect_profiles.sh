#!/bin/bash
# tools/collect_profiles.sh <tag> -- the per-round evidence kept under profiles/ (run on the GPU box: `gpurun -- bash
# tools/collect_profiles.sh r2`; writes gpurun_out/<tag>_profiles/, to be copied into profiles/<tag>_*).  Counter passes
# (--pmc) are separate runs with --kernel-trace only, as the pool requires.
set -u
tag=${1:-rX}
out=gpurun_out/${tag}_profiles
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
./tools/mulbench > "$out/mulbench.txt" 2>&1
./tools/h2bench ntt 24 20 msm 14 254 10 msm 16 254 10 msm 18 254 10 msm 20 254 5 msm 22 254 3 msm 24 254 2 msm 20 16 5 msm 20 1 5 eval 25 3 > "$out/h2bench.txt" 2>&1
./tools/membench > "$out/membench.txt" 2>&1
for L in 16 18 20 22 24; do ./tools/h2bench msmt $L 254 3; done > "$out/msm_table.txt" 2>&1
for H in 8 2; do H2BENCH_HOT=$H ./tools/h2bench msmt 24 254 2; done >> "$out/msm_table.txt" 2>&1
rocprofv3 --kernel-trace -d "$out/p_t24" -o t -- ./tools/h2bench msmt 24 254 2 > /dev/null 2>&1
python3 tools/experiments/split_summary.py "$(find "$out/p_t24" -name '*results.db' | head -1)" > "$out/msm_table_2p24_kernels.txt"
python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -d "$out/p_bench" -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/bench_line_profiled.json" 2>/dev/null
python3 tools/rocprof_summary.py "$(find "$out/p_bench" -name '*results.db' | head -1)" "$out/bench_kernel_stats.txt" > /dev/null
CMD="./tools/h2bench ntt 24 2 msm 20 254 2 eval 25 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/p_f" -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/p_w" -o w -- $CMD > /dev/null 2>&1
python3 tools/hbm_traffic.py "$(find "$out/p_f" -name '*results.db' | head -1)" "$(find "$out/p_w" -name '*results.db' | head -1)" "$CMD" > "$out/hbm_traffic.json"
rocprofv3 --kernel-trace --stats -d "$out/p_k22" -o k22 -- python3 tools/prove_bench.py 22 3 > "$out/create_proof_k22.txt" 2>/dev/null
python3 tools/rocprof_summary.py "$(find "$out/p_k22" -name '*results.db' | head -1)" "$out/create_proof_k22_kernel_stats.txt" > /dev/null
python3 tools/prove_bench.py 24 3 > "$out/create_proof_k24.txt" 2>&1
python3 tools/lookup_bench.py 18 > "$out/lookup_k18.txt" 2>&1
# round 3: the NTT leg alone under the profiler (the k_ntt_pass average the roofline quotes), its SQ counters, the wide circuit
# (--wide-k22 0 too: round 5's run lacked it and its "NTT-only" summary held a whole wide k = 22 proof)
rocprofv3 --kernel-trace --stats -d "$out/p_ntt" -o ntt -- python3 bench.py --steps 20 --warmup 5 --no-msm --prove-k 0 --k24 0 --wide-k 0 --wide-k22 0 --no-cpu-baseline > "$out/bench_ntt_only_line.json" 2>/dev/null
python3 tools/rocprof_summary.py "$(find "$out/p_ntt" -name '*results.db' | head -1)" "$out/bench_ntt_only_kernel_stats.txt" > /dev/null
if grep -q "k_acc_slice\|h2_evalh_gen\|k_msm" "$out/bench_ntt_only_kernel_stats.txt"; then
    echo "collect_profiles.sh: bench_ntt_only_kernel_stats.txt is NOT NTT-only (MSM / evaluate_h kernels in it): a leg was left on" >&2
    mv "$out/bench_ntt_only_kernel_stats.txt" "$out/bench_ntt_only_kernel_stats.CONTAMINATED.txt"
fi
bash tools/experiments/nttpmc.sh > "$out/ntt_pass_pmc.txt" 2>&1
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
python3 tools/wide_bench.py 20 16 > "$out/create_proof_wide_k20.txt" 2>&1
H2_SIDE_INTT=0 python3 tools/wide_bench.py 20 16 - compact >> "$out/create_proof_wide_k20.txt" 2>&1   # phases without the side-stream overlap
python3 tools/wide_bench.py 20 16 coset >> "$out/create_proof_wide_k20.txt" 2>&1
python3 tools/wide_bench.py 22 16 > "$out/create_proof_wide_k22.txt" 2>&1
python3 tools/wide_bench.py 22 16 coset >> "$out/create_proof_wide_k22.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/p_wide" -o w -- python3 tools/wide_bench.py 20 16 > /dev/null 2>&1
python3 tools/rocprof_summary.py "$(find "$out/p_wide" -name '*results.db' | head -1)" "$out/create_proof_wide_k20_kernel_stats.txt" > /dev/null
python3 tools/rocprof_timeline.py "$(find "$out/p_wide" -name '*results.db' | head -1)" 1200 0 > "$out/create_proof_wide_k20_timeline.txt"
# round 5: evaluate_h alone -- the interpreter kernels against the kernels the library generates (csrc/evalh_gen.cpp), with the
# generator's options swept -- and the generated kernel's counters
python3 tools/experiments/evalh_probe.py mini 25 H2_JIT_FACTOR=0 H2_JIT_WAVES=3 > "$out/evalh_probe.txt" 2>&1
python3 tools/experiments/evalh_probe.py wide 22 H2_JIT_FACTOR=0 H2_JIT_WAVES=3 H2_JIT_STAGE_PRODUCTS=60 >> "$out/evalh_probe.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$out/p_e1" -o e -- python3 tools/experiments/evalh_probe.py mini 25 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/p_e2" -o e -- python3 tools/experiments/evalh_probe.py mini 25 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/p_e3" -o e -- python3 tools/experiments/evalh_probe.py mini 25 > /dev/null 2>&1
{ echo "# rocprofv3 --pmc passes over tools/experiments/evalh_probe.py mini 25 (mini-PLONK, 2^25 points): the interpreter kernels"
  echo "# (k_evalh_expr + k_evalh_perm, 4 launches each) and the library-generated kernel h2_evalh_gen (31 launches); per-launch averages"
  for d in p_e1 p_e2 p_e3; do python3 tools/pmc_summary.py "$(find "$out/$d" -name '*results.db' | head -1)" evalh; done; } > "$out/evalh_pmc.txt" 2>&1
python3 tools/evalh_bench.py 20 30 40 > "$out/evalh_gate_sets.txt" 2>&1
python3 tools/evalh_bench.py 20 60 200 >> "$out/evalh_gate_sets.txt" 2>&1
# the multi-rank flow of the default bench, dry-run with four gloo ranks sharing the GPU (communication per phase, replicas leg)
H2_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_line_gloo4_dry_run.json" 2> "$out/bench_gloo4.err"
rm -rf "$out"/p_*
ls -la "$out"
