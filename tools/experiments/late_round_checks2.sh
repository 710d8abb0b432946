#!/bin/bash
# second batch of end-of-round checks: larger random circuits, another MSM seed, more multi-rank circuits
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/late2; mkdir -p $out
H2_FUZZ_KMAX=14 timeout 500 python3 tools/prover_fuzz.py 400 9001 satisfiable > $out/prover_fuzz_k14_sat_seed9001.txt 2>&1
timeout 400 python3 tools/msm_fuzz.py 300 77 tables > $out/msm_fuzz_seed77.txt 2>&1
timeout 700 python3 tools/multirank_fuzz.py 540 2000 > $out/multirank_fuzz_seed2000.txt 2>&1
tail -n 2 $out/*.txt
