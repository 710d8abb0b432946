cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/side_on; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out/pl -o m -- python3 tools/wide_bench.py 20 16 - compact > $out/run.txt 2>/dev/null
db="$(find $out/pl -name '*results.db' | head -1)"
python3 - "$db" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'cop' in t.lower() or 'kern' in t.lower()])
k = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
end = max(r[2] for r in k); t0 = end - 105e6
try:
    cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
    print(cols)
    mc = list(db.execute("select * from memory_copies order by start"))
except Exception as e:
    print("no memory_copies:", e); mc = []
ev = []
for name, s, e, g, w in k:
    if s >= t0: ev.append((s, e, 'K ' + re.sub(r"\(.*", "", name).replace("void ", "").replace("h2::", "")[:30] + " grid %d" % g))
if mc:
    si, ei = cols.index('start'), cols.index('end')
    for r in mc:
        if r[si] >= t0:
            d = dict(zip(cols, r))
            ev.append((r[si], r[ei], 'C %s %s bytes' % (d.get('name', ''), d.get('size', d.get('bytes', '?')))))
ev.sort()
for s, e, n in ev:
    if 28e6 < s - t0 < 44e6 and (e - s > 30e3 or n.startswith('C')): print("%9.1f .. %9.1f %9.1f us %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $out/pl
tail -2 $out/run.txt | cut -c1-250
