cd "$GRAFT_REPO_ROOT"
for G in "" 16 32; do
  echo "== H2_ADVICE_GROUP='$G'"
  if [ -n "$G" ]; then export H2_ADVICE_GROUP=$G; else unset H2_ADVICE_GROUP; fi
  python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-120
  python3 tools/wide_bench.py 22 16 2>&1 | grep "rep 2" | cut -c1-120
done
unset H2_ADVICE_GROUP
python3 tools/wide_bench.py 20 16 - compact 2>&1 | grep "rep 2" | cut -c1-120
python3 tools/lookup_bench.py 18 2>&1 | grep "rep 2" | cut -c1-60
python -m pytest tests/test_gpu_msm_fused.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
