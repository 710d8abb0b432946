#!/bin/bash
# the population of FRESH result pages inside the host-slice calls: MADV_POPULATE_WRITE (H2_HOST_PREFAULT_POPULATE=1) against
# touching every page (the default), 4 / 8 / 16 threads, and none (H2_HOST_PREFAULT=0)
mkdir -p gpurun_out/r6
for rep in 1 2; do
for mode in "1 8" "1 4" "0 8" "0 4" "0 16" "1 0"; do
    set -- $mode
    echo "== H2_HOST_PREFAULT_POPULATE=$1 H2_HOST_PREFAULT=$2"
    H2_HOST_PREFAULT_POPULATE=$1 H2_HOST_PREFAULT=$2 timeout 300 python -u tools/experiments/pageable_calls_probe.py 2>&1 | grep "FRESH\|lincomb"
done
done
