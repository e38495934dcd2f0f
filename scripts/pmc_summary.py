"""Per-kernel HBM-side traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: pmc_summary.py <dir_fetch> <dir_write> <out.json>
Values are KB per launch as the counters report them; FETCH_SIZE must be doubled on gfx950 (the guide's correction;
bench.py applies it)."""
import csv, glob, json, sys, collections
import numpy as np


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return out


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(fe) | set(wr)):
    a, b = np.array(fe.get(k, [0.0])), np.array(wr.get(k, [0.0]))
    res[k] = {"calls": int(len(a)), "fetch_kb_p50": float(np.percentile(a, 50)), "fetch_kb_p90": float(np.percentile(a, 90)),
              "write_kb_p50": float(np.percentile(b, 50)), "write_kb_p90": float(np.percentile(b, 90))}
json.dump(res, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in res.items():
    if v["calls"] > 50:
        print("%-44s n=%6d fetch p90 %9.1f KB  write p90 %9.1f KB" % (k[-44:], v["calls"], v["fetch_kb_p90"], v["write_kb_p90"]))
