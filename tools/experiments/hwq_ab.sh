cd "$GRAFT_REPO_ROOT"
for Q in "" 8 16; do
  echo "== GPU_MAX_HW_QUEUES='$Q'"
  if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
  python3 tools/prove_bench.py 22 3 2>&1 | grep "rep 2" | cut -c1-60
  python3 tools/prove_bench.py 24 3 2>&1 | grep "rep 2" | cut -c1-60
  python3 tools/wide_bench.py 20 16 2>&1 | grep "rep 2" | cut -c1-40
  python3 tools/wide_bench.py 20 16 - compact 2>&1 | grep "rep 2" | cut -c1-40
  python3 tools/wide_bench.py 22 16 2>&1 | grep "rep 2" | cut -c1-40
  python3 tools/lookup_bench.py 18 2>&1 | grep "rep 2" | cut -c1-40
  ./tools/h2bench msmt 20 254 5 2>/dev/null | grep msmt | cut -c60-170
  ./tools/h2bench msmt 24 254 2 2>/dev/null | grep msmt | cut -c60-170
done
