cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc1 -o p -f csv -- $R/tools/h2bench ntt 24 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o p -f csv -- $R/tools/h2bench ntt 24 3 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in ['pmc1','pmc2']:
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv'%d,recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        for k in acc:
            if 'ntt_pass' in k:
                print(d,k)
                for c,v in sorted(acc[k].items()): print('   %-28s %.4g'%(c,v))
PY
