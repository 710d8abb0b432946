for c in 1 2 3; do echo "== NTT columns log $c"; H2_NTT_LOGC=$c ./tools/h2bench ntt 24 20 | cut -c1-80; H2_NTT_LOGC=$c ./tools/h2bench ntt 22 20 | cut -c1-80; done
for s in 4 5 6 7; do echo "== MSM slice log $s"; H2_MSM_SLICE_LOG=$s ./tools/h2bench msmt 24 254 2 | grep msmt | cut -c1-150; H2_MSM_SLICE_LOG=$s ./tools/h2bench msmt 20 254 5 | grep msmt | cut -c1-150; done
echo "== default"; ./tools/h2bench msmt 24 254 2 | grep msmt | cut -c1-150; ./tools/h2bench msmt 20 254 5 | grep msmt | cut -c1-150
for q in 16 32 64; do echo "== reduce qm $q"; H2_MSM_REDUCE_QM=$q ./tools/h2bench msmt 24 254 2 | grep msmt | cut -c1-150; H2_MSM_REDUCE_QM=$q ./tools/h2bench msmt 20 254 5 | grep msmt | cut -c1-150; done
