#!/bin/bash
# end-of-round checks on the final build: long fuzz runs with fresh seeds, the bare `--gpus N` launcher at 4 and 8 gloo
# ranks on one GPU (small sizes), the lookup-circuit timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/late; mkdir -p $out
timeout 420 python3 tools/msm_fuzz.py 300 41 tables > $out/msm_fuzz_seed41.txt 2>&1
timeout 420 python3 tools/prover_fuzz.py 300 5001 > $out/prover_fuzz_seed5001.txt 2>&1
timeout 300 python3 tools/prover_fuzz.py 180 7001 satisfiable > $out/prover_fuzz_sat_seed7001.txt 2>&1
for n in 4 8; do
  H2_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus $n --steps 2 --warmup 1 --k24 0 --prove-k 14 --wide-k 12 --log-n 18 --no-cpu-baseline > $out/bench_gloo_$n.json 2> $out/bench_gloo_$n.err
  echo "gloo ranks $n rc $?" >> $out/launcher.txt
done
bash tools/experiments/timeline_lookup.sh 18 700
