#!/bin/bash
# single / batched MSM over a shifted-base table against the number of digits (= bucket count) at mid sizes: is the
# default (table_default_digits) where the latency-bound tails say it should be?  H2_MSM_TABLE_FORCE lifts the rule that a
# table of more digits than the windowed shape has windows is not used.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/msm_digits_sweep.txt; : > $out
for L in ${LOGS:-16 18 20}; do
  echo "== log_n $L default" >> $out
  ./tools/h2bench msmt $L 254 5 2>/dev/null | grep msmt >> $out
  for D in ${DIGITS:-16 17 18 19 20 22 24 26}; do
    echo "-- digits $D" >> $out
    H2_MSM_TABLE_FORCE=1 H2_MSM_TABLE_DIGITS=$D ./tools/h2bench msmt $L 254 5 2>/dev/null | grep msmt >> $out
  done
done
cat $out
