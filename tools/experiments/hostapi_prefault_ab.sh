# the literal drop-in's k = 22 proof (bench.py create_proof.host_slice_api) with the library populating the pages of result
# vectors in ordinary memory before the copy back (huge pages advised): touching every page from four threads (the default),
# MADV_POPULATE_WRITE from eight (H2_HOST_PREFAULT_POPULATE=1 H2_HOST_PREFAULT=8: the form before), and not at all
# (H2_HOST_PREFAULT=0: the runtime's staging thread takes the first-touch faults).  Same box, alternating.
for round in 1 2 3; do
  for P in "0 4" "1 8" "0 0"; do
    set -- $P
    H2_HOST_PREFAULT_POPULATE=$1 H2_HOST_PREFAULT=$2 python3 bench.py --no-msm --wide-k 0 --wide-k22 0 --k24 0 --cpu-prove-k 0 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['create_proof']['host_slice_api']
print('POPULATE=$1 threads=$2  pageable %.3f s (in library calls %.3f)  pinned %.3f s (%.3f)  pageable phases %s' % (h['pageable']['seconds'], h['pageable']['seconds_inside_library_calls'], h['pinned']['seconds'], h['pinned']['seconds_inside_library_calls'], h['pageable']['phases_ms']))"
  done
done
