#!/bin/bash
# the bare launcher at 8 gloo ranks on one GPU, repeated: is the wide-circuit leg's result stable?
# usage: bench8_repeat.sh <tag> <runs> [bench args...]   (environment knobs are inherited)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/b8; mkdir -p $out
tag=$1; runs=$2; shift 2
for i in $(seq 1 $runs); do
  H2_BENCH_BACKEND=gloo timeout 400 python3 bench.py --gpus 8 --steps 2 --warmup 1 --k24 0 --prove-k 0 --wide-k 12 --log-n 18 --no-cpu-baseline "$@" > $out/${tag}_$i.json 2> $out/${tag}_$i.err
  python3 - <<PY
import json
try:
    d=json.loads(open('$out/${tag}_$i.json').read().strip().splitlines()[-1])
    w=d.get('create_proof_wide',{})
    print('$tag run $i', w.get('error'), ((w.get('sharded') or {}).get('proof_sha256') or '')[:16], w.get('verified'))
except Exception as e:
    print('$tag run $i no line', e)
PY
done
