"""Per-kernel medians of SQ counters from a rocprofv3 --pmc run (counter_collection.csv under <dir>).
Usage: pmc_sq_summary.py <dir> [kernel-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; pats = sys.argv[2:]
rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pats and not any(p in k for p in pats):
            continue
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    print(k[:110])
    med = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    for c, v in sorted(med.items()):
        print("   %-26s %16.0f  (%d launches)" % (c, v, len(cs[c])))
    if "SQ_LDS_IDX_ACTIVE" in med and med["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in med:
        print("   bank-conflict share of LDS cycles: %.2f" % (med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_WAVE_CYCLES" in med and med["SQ_WAVE_CYCLES"] > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in med:
                print("   %-26s / wave cycles = %.2f" % (c, med[c] / med["SQ_WAVE_CYCLES"]))
