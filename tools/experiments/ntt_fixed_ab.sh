#!/bin/bash
# same-box A/B of the fixed-geometry instance of k_ntt_pass (B = 8, 4 columns: compile-time strides) against the generic one
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
  for f in 1 0; do
    echo "== H2_NTT_FIXED=$f"
    for a in "24 20" "25 10" "22 20" "20 40"; do H2_NTT_FIXED=$f ./tools/h2bench ntt $a | grep -v amdgpu; done
  done
done
