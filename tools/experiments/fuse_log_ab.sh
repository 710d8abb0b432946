cd "$GRAFT_REPO_ROOT"
for F in 22 23 24 25; do
  echo "== H2_MSM_FUSE_LOG=$F"
  H2_MSM_FUSE_LOG=$F python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-120
  H2_MSM_FUSE_LOG=$F python3 tools/wide_bench.py 20 16 - compact 2>&1 | grep "rep 2" | cut -c1-120
  H2_MSM_FUSE_LOG=$F python3 tools/prove_bench.py 22 3 2>&1 | grep "rep 2" | cut -c1-100
done
