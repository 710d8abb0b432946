cd "$GRAFT_REPO_ROOT"
python3 -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p, '->', s.priority)
    except Exception as e: print(p, 'error', e)
"
for P in "" "-1,0" "0,1" "-1,1" "1,-1"; do
  echo "== H2_STREAM_PRIORITY='$P'"
  H2_STREAM_PRIORITY="$P" python3 tools/wide_bench.py 20 16 - compact 2>/dev/null | grep "rep [12]" | cut -c1-330
  H2_STREAM_PRIORITY="$P" python3 tools/lookup_bench.py 18 2>/dev/null | grep "rep 2" | cut -c1-200
done
