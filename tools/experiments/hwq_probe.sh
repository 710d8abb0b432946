cd "$GRAFT_REPO_ROOT"
for Q in "" 2 8 16; do
  for A in 2 8; do
    echo "== GPU_MAX_HW_QUEUES='$Q' H2_ADVICE_AHEAD=$A"
    if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
    H2_ADVICE_AHEAD=$A python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-260
  done
done
