cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/busy24; mkdir -p $out
rocprofv3 --kernel-trace -d $out/pl -o m -- python3 tools/prove_bench.py ${1:-24} 3 > $out/prove.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/pl -name '*results.db' | head -1)" 100000 0 > $out/timeline.txt
rm -rf $out/pl
python3 - <<'PY'
import re
rows=[]
for line in open('gpurun_out/busy24/timeline.txt'):
    m=re.match(r'\s*([\d.]+) us\s+\+\s*(-?[\d.]+) gap\s+([\d.]+) us\s+(\S+)',line)
    if m: rows.append((float(m.group(1)),float(m.group(3)),m.group(4).replace('h2::','')))
end=max(a+b for a,b,_ in rows)
import sys
span=float(sys.argv[1]) if len(sys.argv)>1 else 190000.0
sel=[r for r in rows if r[0]>=end-span]
iv=sorted((a,a+b,n) for a,b,n in sel)
busy=0; cs,ce=iv[0][0],iv[0][1]; gaps=[]; last=iv[0][2]
for a,b,n in iv[1:]:
    if a>ce:
        busy+=ce-cs; gaps.append((a-ce,ce-(end-span),last,n)); cs,ce=a,b
    else: ce=max(ce,b)
    if b>=ce: last=n
busy+=ce-cs
print("last %.0f ms: busy %.1f ms (%.0f%%), %d gaps"%(span/1e3,busy/1e3,100*busy/span,len(gaps)))
for g in sorted(gaps,reverse=True)[:25]: print("%8.1f us at %9.1f after %-24s before %s"%g)
PY
tail -3 $out/prove.txt
