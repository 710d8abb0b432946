# k_reduce of the windowed shapes: the bound on its workgroups (H2_MSM_REDUCE_WGS; 256 = one per CU) -- shorter chains per quad
# against two workgroups sharing a CU
cd "$GRAFT_REPO_ROOT"
for W in 256 512 1024; do
  echo "== H2_MSM_REDUCE_WGS=$W"
  export H2_MSM_REDUCE_WGS=$W
  H2_SIDE_INTT=0 python3 tools/wide_bench.py 20 16 - compact 2>&1 | grep "rep 2" | cut -c1-130
  python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-130
  ./tools/h2bench msm 20 254 5 msm 20 16 5 msm 16 254 5 2>/dev/null | grep "^msm" | cut -c1-150
done
