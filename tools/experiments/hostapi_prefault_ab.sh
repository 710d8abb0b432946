# the literal drop-in's k = 22 proof (bench.py create_proof.host_slice_api) with the library populating the pages of result
# vectors in ordinary memory by eight threads (+ MADV_HUGEPAGE) before the copy back -- H2_HOST_PREFAULT=8, the default -- and
# without (H2_HOST_PREFAULT=0: the runtime's staging thread takes the first-touch faults).  Same box, alternating.
for round in 1 2 3; do
  for P in 8 0; do
    H2_HOST_PREFAULT=$P python3 bench.py --no-msm --wide-k 0 --wide-k22 0 --k24 0 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['create_proof']['host_slice_api']
print('H2_HOST_PREFAULT=$P  pageable %.3f s  pinned %.3f s  pageable phases %s' % (h['pageable']['seconds'], h['pinned']['seconds'], h['pageable']['phases_ms']))"
  done
done
