# GPU-busy fraction and the largest idle gaps of the LAST <span_us> microseconds of a command's kernel trace:
#   bash tools/experiments/busy.sh <span_us> <from_us> <to_us> python3 tools/wide_bench.py 20 16 - compact
# (<from_us> <to_us>: also print the kernel time by name inside that window of the span; 0 0 = skip)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
span=$1; from=$2; to=$3; shift 3
out=gpurun_out/busy; mkdir -p $out
rocprofv3 --kernel-trace -d $out/pl -o m -- "$@" > $out/run.txt 2>/dev/null
python3 tools/rocprof_timeline.py "$(find $out/pl -name '*results.db' | head -1)" 1000000 0 > $out/timeline.txt
rm -rf $out/pl
python3 - $span $from $to <<'PY'
import re, sys, collections
span, lo, hi = float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
rows = []
for line in open('gpurun_out/busy/timeline.txt'):
    m = re.match(r'\s*([\d.]+) us\s+\+\s*(-?[\d.]+) gap\s+([\d.]+) us\s+(\S+)', line)
    if m: rows.append((float(m.group(1)), float(m.group(3)), m.group(4).replace('h2::', '')))
end = max(a + b for a, b, _ in rows)
t0 = end - span
iv = sorted((a, a + b, n) for a, b, n in rows if a >= t0)
busy = 0; cs, ce = iv[0][0], iv[0][1]; gaps = []; last = iv[0][2]
for a, b, n in iv[1:]:
    if a > ce:
        busy += ce - cs; gaps.append((a - ce, ce - t0, last, n)); cs, ce = a, b
    else: ce = max(ce, b)
    if b >= ce: last = n
busy += ce - cs
print("last %.1f ms: busy %.1f ms (%.0f%%), %d gaps, %.1f ms idle" % (span / 1e3, busy / 1e3, 100 * busy / span, len(gaps), sum(g[0] for g in gaps) / 1e3))
for g in sorted(gaps, reverse=True)[:12]: print("%8.1f us at %9.1f after %-24s before %s" % g)
if hi > lo:
    c = collections.Counter(); n = collections.Counter()
    for a, b, k in iv:
        if lo <= a - t0 < hi: c[k] += b - a; n[k] += 1
    print("kernel time by name in [%.0f, %.0f) us of the span:" % (lo, hi))
    for k, v in c.most_common(16): print("  %-34s %9.1f us %5d launches" % (k, v, n[k]))
PY
tail -4 $out/run.txt | cut -c1-400
