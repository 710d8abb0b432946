# per-pass durations and VALU counters of k_ntt_pass at 2^24 (three 8-bit passes; launch order mod 3), with the tabulated twiddles
# as (plain, quotient) pairs (H2_NTT_CONSTW=1) and in Montgomery form (=0).  Separate rocprofv3 runs: kernel trace; --pmc.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for CW in 1 0; do
  export H2_NTT_CONSTW=$CW
  rm -rf $R/gpurun_out/pt_t$CW $R/gpurun_out/pt_c$CW
  rocprofv3 --kernel-trace -d $R/gpurun_out/pt_t$CW -o p -f csv -- $R/tools/h2bench ntt 24 10 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/pt_c$CW -o p -f csv -- $R/tools/h2bench ntt 24 3 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for cw in (1, 0):
    print("== H2_NTT_CONSTW=%d" % cw)
    for f in glob.glob('gpurun_out/pt_t%d/**/*kernel_trace.csv' % cw, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'k_ntt_pass' in r['Kernel_Name']]
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        rows = rows[len(rows) // 2 // 3 * 3:]                      # the second half: the steady clock
        acc = collections.defaultdict(list)
        for i, r in enumerate(rows):
            acc[i % 3].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        r0 = rows[0]
        print('  %s  lds=%s scratch=%s vgpr=%s wg=%s' % (r0['Kernel_Name'][:60], r0.get('LDS_Block_Size'), r0.get('Scratch_Size'), r0.get('VGPR_Count'), r0.get('Workgroup_Size_X', r0.get('Workgroup_Size'))))
        for p in sorted(acc):
            v = sorted(acc[p])
            print('  pass %d: median %.1f us  mean %.1f us over %d launches' % (p, v[len(v) // 2], sum(v) / len(v), len(v)))
        print('  sum of medians: %.1f us per transform' % sum(sorted(acc[p])[len(acc[p]) // 2] for p in acc))
    for f in glob.glob('gpurun_out/pt_c%d/**/*counter_collection.csv' % cw, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'k_ntt_pass' in r['Kernel_Name']]
        ids = sorted({int(r['Dispatch_Id']) for r in rows})
        pos = {d_: i % 3 for i, d_ in enumerate(ids)}
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:
            acc[pos[int(r['Dispatch_Id'])]][r['Counter_Name']] += float(r['Counter_Value'])
        n = len(ids) // 3
        for p in sorted(acc):
            a = acc[p]
            print('  pass %d: VALU instr/element %.0f  LDS instr/element %.1f  cycles per VALU instr per SIMD %.2f  waves resident per SIMD %.2f' % (
                p, a['SQ_INSTS_VALU'] / n * 64 / 2**24, a['SQ_INSTS_LDS'] / n * 64 / 2**24,
                a['GRBM_GUI_ACTIVE'] / n / 8 / (a['SQ_INSTS_VALU'] / n / 1024), a['SQ_WAVE_CYCLES'] / max(a['SQ_BUSY_CYCLES'], 1) / 4 * 4))
PY
rm -rf gpurun_out/pt_t1 gpurun_out/pt_t0 gpurun_out/pt_c1 gpurun_out/pt_c0
