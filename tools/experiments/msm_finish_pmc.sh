# SQ / cache counters of k_acc_slice for one 2^24 MSM over a table (per launch); two --pmc runs (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-24}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcA -o p -f csv -- $R/tools/h2bench msmt $L 254 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmcB -o p -f csv -- $R/tools/h2bench msmt $L 254 2 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ['pmcA', 'pmcB']:
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-28:]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); ids[k].add(r['Dispatch_Id'])
        for k in acc:
            if 'k_finish' in k or 'acc_slice' in k or 'k_reduce' in k:
                n = len(ids[k])
                print(d, k, 'launches', n, ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(acc[k].items())))
PY
rm -rf gpurun_out/pmcA gpurun_out/pmcB
