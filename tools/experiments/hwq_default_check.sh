cd "$GRAFT_REPO_ROOT"
unset GPU_MAX_HW_QUEUES
python3 tools/wide_bench.py 22 16 2>&1 | grep "rep 2" | cut -c1-40
python3 tools/wide_bench.py 22 16 - compact 2>&1 | grep "rep 2" | cut -c1-40
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_hwq.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/bench_hwq.json')); print(json.dumps(d['config']['headline']))"
GPU_MAX_HW_QUEUES=4 python3 tools/wide_bench.py 22 16 2>&1 | grep "rep 2" | cut -c1-40
