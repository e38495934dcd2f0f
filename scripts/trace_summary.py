"""Summarise a rocprofv3 kernel trace: per-kernel duration percentiles, and -- for kernels that are launched on a budget and
leave at once when their solve has converged -- the split into WORKING launches and launches that found nothing to do
(duration < 60 % of the kernel's 95th percentile).

    python scripts/trace_summary.py <rocprof output dir> [--last FRACTION] [--min-calls N]

--last 0.33: only the last third of the trace's time line (steady state: launch budgets settled, graphs captured)."""
import csv, glob, sys, collections
import numpy as np
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--min-calls"]
last = 1.0
if "--last" in sys.argv:
    last = float(sys.argv[sys.argv.index("--last") + 1])
    args = [a for a in args if a != sys.argv[sys.argv.index("--last") + 1]]
f = glob.glob(args[0] + '/*/*kernel_trace.csv')[0]
minc = 20
if "--min-calls" in sys.argv:
    minc = int(sys.argv[sys.argv.index("--min-calls") + 1])
rows = list(csv.DictReader(open(f)))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
cut = t1 - last * (t1 - t0)
rows = [r for r in rows if int(r['Start_Timestamp']) >= cut]
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'].split('(')[0][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
span = (max(int(r['End_Timestamp']) for r in rows) - min(int(r['Start_Timestamp']) for r in rows)) / 1e3
print("window: last %.0f %% of the trace, %.1f ms of time line, %.1f ms of kernel time (%.0f %% busy)" % (100 * last, span / 1e3, tot / 1e3, 100 * tot / span))
print("%-46s %7s %9s %6s %7s %7s %7s | %7s %8s %7s %8s" % ("kernel", "calls", "total_us", "%", "p10", "p50", "p90", "working", "mean_us", "no-op", "mean_us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    if len(v) < minc:
        continue
    thr = 0.6 * np.percentile(v, 95)
    w, n = v[v >= thr], v[v < thr]
    print("%-46s %7d %9.0f %6.1f %7.2f %7.2f %7.2f | %7d %8.2f %7d %8.2f" % (k, len(v), v.sum(), 100 * v.sum() / tot, *np.percentile(v, [10, 50, 90]),
          len(w), w.mean() if len(w) else 0.0, len(n), n.mean() if len(n) else 0.0))
