# columns per NTT tile (H2_NTT_LOGC: 2^c columns x 256 rows; default 2) with the lazy-domain kernel
for round in 1 2; do
  for c in 2 1 3; do
    echo "== H2_NTT_LOGC=$c"
    H2_NTT_LOGC=$c ./tools/h2bench ntt 24 20 ntt 22 20 | cut -c1-110
  done
done
