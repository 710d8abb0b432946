cd "$GRAFT_REPO_ROOT"
H2_PROVER_HOST_TRACE=1 python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "host trace\|rep [12]" | cut -c1-900 | tail -3
python3 tools/wide_bench.py 20 16 - compact 2>&1 | grep "rep [12]" | cut -c1-120
python3 tools/wide_bench.py 20 16 2>&1 | grep "rep [12]" | cut -c1-120
python3 tools/wide_bench.py 22 16 2>&1 | grep "rep [12]" | cut -c1-120
python3 tools/prove_bench.py 22 3 2>&1 | grep "rep [12]" | cut -c1-120
python3 tools/prove_bench.py 24 3 2>&1 | grep "rep [12]" | cut -c1-120
