# same box, same library: the lazy domain of the NTT stage loops (H2_NTT_LAZY, default on) against canonical residues
python -m pytest tests/test_gpu_parity.py tests/test_gpu_numerics.py -x -q -m gpu -k "ntt or coset or commit_lagrange or intt" 2>&1 | tail -1
for round in 1 2 3; do
  for lazy in 1 0; do
    echo "== H2_NTT_LAZY=$lazy"
    H2_NTT_LAZY=$lazy ./tools/h2bench ntt 24 20 ntt 25 10 ntt 22 20 ntt 20 40 ntt 16 100 | cut -c1-120
  done
done
