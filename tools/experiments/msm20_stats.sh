# kernel shares of a single and a batched MSM over a table (h2bench msmt L 254 reps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=${1:-20}
rocprofv3 --kernel-trace --stats -d gpurun_out/msm_st -o m -- ./tools/h2bench msmt $L 254 5 > gpurun_out/msm${L}_run.txt 2>&1
python3 tools/rocprof_summary.py "$(find gpurun_out/msm_st -name '*results.db' | head -1)" gpurun_out/msm${L}_kernel_stats.txt > /dev/null
rm -rf gpurun_out/msm_st
head -30 gpurun_out/msm${L}_kernel_stats.txt
