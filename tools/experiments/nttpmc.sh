# SQ counters of k_ntt_pass at 2^24 (three 8-bit passes), per launch and per pass position (launch order mod 3).
# Two separate rocprofv3 runs (--pmc with --kernel-trace only, as the pool requires).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc1 -o p -f csv -- $R/tools/h2bench ntt 24 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o p -f csv -- $R/tools/h2bench ntt 24 3 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ['pmc1', 'pmc2']:
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'k_ntt_pass' in r['Kernel_Name']]
        ids = sorted({int(r['Dispatch_Id']) for r in rows})
        pos = {d_: i % 3 for i, d_ in enumerate(ids)}          # forward and inverse transforms alternate: passes 0, 1, 2
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:
            acc[pos[int(r['Dispatch_Id'])]][r['Counter_Name']] += float(r['Counter_Value'])
        n = len(ids) // 3
        print('%s: %d launches of %s' % (d, len(ids), rows[0]['Kernel_Name'][:48]))
        for p in sorted(acc):
            print('  pass %d (per launch):' % p, '  '.join('%s=%.4g' % (c, v / n) for c, v in sorted(acc[p].items())))
            if 'SQ_INSTS_VALU' in acc[p]:
                print('     VALU instructions per element: %.0f   cycles per VALU instruction per SIMD: %.2f   (GRBM_GUI_ACTIVE is summed over the 8 XCDs)' % (
                    acc[p]['SQ_INSTS_VALU'] / n * 64 / 2**24, acc[p]['GRBM_GUI_ACTIVE'] / n / 8 / (acc[p]['SQ_INSTS_VALU'] / n / 1024)))
PY
