python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or coset or fft or intt" 2>&1 | tail -2
for i in 1 2; do
echo "== table"; ./tools/h2bench ntt 24 20; ./tools/h2bench ntt 25 10; ./tools/h2bench ntt 22 20; ./tools/h2bench ntt 20 20; ./tools/h2bench ntt 18 20
echo "== no table"; H2_NTT_LAST_TABLE=0 ./tools/h2bench ntt 24 20; H2_NTT_LAST_TABLE=0 ./tools/h2bench ntt 25 10; H2_NTT_LAST_TABLE=0 ./tools/h2bench ntt 22 20; H2_NTT_LAST_TABLE=0 ./tools/h2bench ntt 20 20; H2_NTT_LAST_TABLE=0 ./tools/h2bench ntt 18 20
done
