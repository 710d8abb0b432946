#!/bin/bash
# the MSM leg alone at 8 gloo ranks on one GPU: which assertion fires, how often?   usage: bench8_msm.sh <tag> <runs> [bench args]
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/b8m; mkdir -p $out
tag=$1; runs=$2; shift 2
export H2_BENCH_RANK_GRACE_S=3
for i in $(seq 1 $runs); do
  H2_BENCH_BACKEND=gloo timeout 200 python3 bench.py --gpus 8 --steps 2 --warmup 1 --k24 0 --prove-k 0 --wide-k 0 --log-n 18 --no-cpu-baseline "$@" > $out/${tag}_$i.json 2> $out/${tag}_$i.err
  echo "$tag run $i rc $? :" $(grep -h "leg .* failed" $out/${tag}_$i.err | grep -v "Connection closed" | cut -c1-160 | tr '\n' ';')
done
