# GPU_MAX_HW_QUEUES A/B (the runtime's default is 4; the package / library default is 8): LEGS="wide22 wide22c k22 k24 wide20 msm"
cd "$GRAFT_REPO_ROOT"
for Q in ${QUEUES:-4 8 16}; do
  echo "== GPU_MAX_HW_QUEUES=$Q"
  export GPU_MAX_HW_QUEUES=$Q
  for leg in ${LEGS:-wide22 wide22c k22 k24 wide20 msm}; do
    case $leg in
      wide22)  python3 tools/wide_bench.py 22 16 2>&1 | grep "rep 2" | cut -c1-150 | sed 's/^/wide k=22          /';;
      wide22c) python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-150 | sed 's/^/wide k=22 compact  /';;
      wide20)  python3 tools/wide_bench.py 20 16 2>&1 | grep "rep 2" | cut -c1-150 | sed 's/^/wide k=20          /';;
      k22)     python3 tools/prove_bench.py 22 3 2>&1 | grep "rep 2" | cut -c1-110 | sed 's/^/mini-PLONK k=22    /';;
      k24)     python3 tools/prove_bench.py 24 3 2>&1 | grep "rep 2" | cut -c1-110 | sed 's/^/mini-PLONK k=24    /';;
      msm)     ./tools/h2bench msmt 20 254 5 2>/dev/null | grep msmt | cut -c1-170;;
    esac
  done
done
