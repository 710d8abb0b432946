#!/bin/bash
# the mini-PLONK k = 22 drop-in leg from ORDINARY memory: the runtime's pageable copies (H2_HOST_COPY_THREADS=0) against the library's
# staged path with 2 / 3 threads per transfer, alternating on one box
for round in 1 2 3; do
  for T in 0 2 3; do
    H2_HOST_COPY_THREADS=$T python bench.py --no-msm --no-cpu-baseline --wide-k 0 --wide-k22 0 --k24 0 --cpu-prove-k 0 --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['create_proof']['host_slice_api']
print('H2_HOST_COPY_THREADS=$T  pageable %.3f s (in library calls %.3f)  pinned %.3f s  pageable phases %s' % (h['pageable']['seconds'], h['pageable']['seconds_inside_library_calls'], h['pinned']['seconds'], h['pageable']['phases_ms']))"
  done
done
