"""Summarise a rocprofv3 kernel trace: per-kernel duration percentiles (early-exit vs full launches)."""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'].split('(')[0][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
print("%-46s %7s %9s %6s %6s %6s %6s" % ("kernel", "calls", "total_us", "p10", "p50", "p90", "%"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    if len(v) > 100:
        print("%-46s %7d %9.0f %6.2f %6.2f %6.2f %6.1f" % (k, len(v), v.sum(), *np.percentile(v, [10, 50, 90]), 100 * v.sum() / tot))
