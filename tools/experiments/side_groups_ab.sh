# A/B of the per-group side-stream transforms of the witness columns (prover.py: H2_SIDE_GROUPS=1, the default above k = 20: each
# commitment group's columns go to coefficient form and to the extended domain on the side stream right behind the group's
# commitment, under the later groups' PCIe transfers) against the form of rounds 2-5 (H2_SIDE_GROUPS=0: after the whole phase, on the
# compute stream).  Same box, alternating.  usage: bash tools/experiments/side_groups_ab.sh
for round in 1 2; do
  for G in 1 0; do
    echo "== H2_SIDE_GROUPS=$G"
    H2_SIDE_GROUPS=$G python3 tools/wide_bench.py 22 16 2>&1 | grep "rep [12]" | cut -c1-400
    H2_SIDE_GROUPS=$G python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep [12]" | cut -c1-60
    H2_SIDE_GROUPS=$G python3 tools/prove_bench.py 22 4 2>&1 | grep "rep [23]" | cut -c1-60
    H2_SIDE_GROUPS=$G python3 tools/prove_bench.py 24 3 2>&1 | grep "rep [12]" | cut -c1-60
  done
done
