# k_reduce_planes: workgroups over all planes (H2_MSM_PLANE_WGS; 256 = one per CU) against the single-MSM latency
cd "$GRAFT_REPO_ROOT"
for W in ${WGS:-256 512 768 1024 1536}; do
  echo "== H2_MSM_PLANE_WGS=$W"
  for L in 16 18 20 22; do H2_MSM_PLANE_WGS=$W ./tools/h2bench msmt $L 254 5 2>/dev/null | grep msmt | cut -c1-160; done
done
