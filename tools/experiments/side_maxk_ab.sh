cd "$GRAFT_REPO_ROOT"
for M in 20 24; do
  echo "== H2_SIDE_INTT_MAX_K=$M"
  H2_SIDE_INTT_MAX_K=$M python3 tools/prove_bench.py 22 3 2>&1 | grep "rep 2" | cut -c1-60
  H2_SIDE_INTT_MAX_K=$M python3 tools/prove_bench.py 24 3 2>&1 | grep "rep 2" | cut -c1-60
  H2_SIDE_INTT_MAX_K=$M python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-40
done
