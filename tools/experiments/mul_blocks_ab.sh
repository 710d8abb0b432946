#!/bin/bash
# same-box A/B: the multiplier as one asm statement per instruction (old: tools/experiments/oldlib/libhalo2_hip.so) against
# the assembly blocks of tools/gen_fp_mul.py's `lower` (new: the tree's library).  The old library: a copy of csrc/ with the
# fp_mul_gen.hpp of commit 18add17, `make`, the .so into tools/experiments/oldlib/ (not kept in the tree).
OLD=tools/experiments/oldlib
for round in 1 2; do
  for lib in new old; do
    echo "== $lib"
    if [ $lib = old ]; then export LD_LIBRARY_PATH=$OLD; else unset LD_LIBRARY_PATH; fi
    ./tools/h2bench ntt 24 10 ntt 25 5 ntt 18 50 ntt 14 100 2>/dev/null | grep "^ntt" | cut -c1-100
    ./tools/h2bench msm 14 254 10 msm 16 254 10 msm 18 254 10 msm 20 254 5 2>/dev/null | grep "^msm" | cut -c1-150
    for L in 16 18 20 22 24; do ./tools/h2bench msmt $L 254 3 2>/dev/null | grep msmt | cut -c1-150; done
  done
done
unset LD_LIBRARY_PATH
