# the mini-PLONK proof at k = $1 (default 24): bash tools/experiments/busy24.sh 24 185000   (see busy.sh)
export H2_PROVE_BENCH_NO_TIMINGS=1
exec bash "$(dirname "$0")/busy.sh" ${2:-185000} 0 0 python3 tools/prove_bench.py ${1:-24} 3
