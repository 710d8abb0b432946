# same-box A/B of the NTT passes with their tabulated twiddles as (plain value, quotient) pairs multiplied by fp_mul_const
# (H2_NTT_CONSTW=1, the default) against Montgomery-form tables and fp_mul_wide (H2_NTT_CONSTW=0): one library, the knob is
# read when a plan is built (each h2bench run is a process of its own).  usage: bash tools/experiments/ntt_constw_ab.sh
for round in 1 2 3; do
  echo "== const-operand twiddles (H2_NTT_CONSTW=1)"; H2_NTT_CONSTW=1 ./tools/h2bench ntt 24 20 ntt 25 10 ntt 22 20 ntt 20 20
  echo "== Montgomery twiddles    (H2_NTT_CONSTW=0)"; H2_NTT_CONSTW=0 ./tools/h2bench ntt 24 20 ntt 25 10 ntt 22 20 ntt 20 20
done
