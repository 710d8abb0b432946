#!/bin/bash
# copies from / into ORDINARY host memory that is new to the runtime (torch / malloc allocations: 4 KiB pages, no huge-page advice),
# three ways: the runtime's pageable path with long copies serialised (default), unserialised (H2_HOST_SERIAL_COPIES=0), and the
# library's staged path (H2_HOST_COPY_THREADS=n): h2_intt on 32 MiB vectors, the 64-column proof, the mini-PLONK drop-in leg
mkdir -p gpurun_out/r6
for mode in "H2_HOST_COPY_THREADS=0" "H2_HOST_COPY_THREADS=0 H2_HOST_SERIAL_COPIES=0" "H2_HOST_COPY_THREADS=2" "H2_HOST_COPY_THREADS=4"; do
    echo "== $mode"
    env $mode python tools/experiments/pageable_intt_probe.py 2>&1 | grep "torch clone\|those again\|4 threads\|again"
    env $mode python tools/experiments/hostapi_wide.py 20 pageable 2>&1 | grep "host-slice (pageable)"
    env $mode python bench.py --no-msm --no-cpu-baseline --wide-k 0 --wide-k22 0 --k24 0 --cpu-prove-k 0 --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['create_proof']['host_slice_api']
print('mini-PLONK k = 22: pageable %.3f s (in library calls %.3f)  pinned %.3f s  pageable phases %s' % (h['pageable']['seconds'], h['pageable']['seconds_inside_library_calls'], h['pinned']['seconds'], h['pageable']['phases_ms']))"
done
