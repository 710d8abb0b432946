cd "$GRAFT_REPO_ROOT"
unset GPU_MAX_HW_QUEUES
echo "A: set in-process before import torch"
python3 -c "
import os, sys, runpy
os.environ['GPU_MAX_HW_QUEUES']='8'
sys.argv=['tools/wide_bench.py','22','16']
runpy.run_path('tools/wide_bench.py', run_name='__main__')" 2>&1 | grep "rep 2" | cut -c1-40
echo "B: libamdhip64 loaded (no call), then set, then torch"
python3 -c "
import os, sys, runpy, ctypes, glob
import importlib.util
tl=os.path.join(os.path.dirname(importlib.util.find_spec('torch').origin),'lib','libamdhip64.so')
ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
os.environ['GPU_MAX_HW_QUEUES']='8'
sys.argv=['tools/wide_bench.py','22','16']
runpy.run_path('tools/wide_bench.py', run_name='__main__')" 2>&1 | grep "rep 2" | cut -c1-40
echo "C: import torch, then set, then first HIP call"
python3 -c "
import os, sys, runpy
import torch
os.environ['GPU_MAX_HW_QUEUES']='8'
sys.argv=['tools/wide_bench.py','22','16']
runpy.run_path('tools/wide_bench.py', run_name='__main__')" 2>&1 | grep "rep 2" | cut -c1-40
