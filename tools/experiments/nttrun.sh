set -x
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or coset or fft or intt" 2>&1 | tail -3
for i in 1 2; do ./tools/h2bench ntt 24 20; done
./tools/h2bench ntt 25 10; ./tools/h2bench ntt 22 20; ./tools/h2bench ntt 20 20; ./tools/h2bench ntt 16 20
H2_NTT_DBG=1 ./tools/h2bench ntt 24 20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/nttp -o ntt -- $GRAFT_REPO_ROOT/tools/h2bench ntt 24 6 > /dev/null 2>&1
H2_NTT_DBG=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/nttp_dbg -o ntt -- $GRAFT_REPO_ROOT/tools/h2bench ntt 24 6 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
for d in ['nttp','nttp_dbg']:
    f=glob.glob('gpurun_out/%s/**/*kernel_trace.csv'%d,recursive=True)
    if not f: print(d,'no trace'); continue
    rows=[r for r in csv.DictReader(open(f[0])) if 'k_ntt_pass' in r['Kernel_Name']]
    d_=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
    print(d, len(d_), ' '.join('%.0f'%x for x in d_[-18:]))
PY
