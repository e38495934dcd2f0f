"""Per-kernel table of the profiled bench run: algorithmic bytes per launch (nekstab_amd/roofline.py), launch
duration (rocprofv3 kernel trace, p90 = launches that do full work, p10 = launches that find their solve converged), the
resulting TB/s, and the HBM-side bytes from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE).  Also writes
profiles-style rNN_pmc_traffic.json (bytes per launch keyed by kernel, stamped with the library's source hash).
Usage: kernel_table.py <trace_dir> <pmc_summary.json> <bench_line.json>"""
import collections, csv, glob, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import roofline
trace_dir, pmc_json, bench_json = sys.argv[1:4]
rows = list(csv.DictReader(open(glob.glob(trace_dir + '/*/*kernel_trace.csv')[0])))
dur = collections.defaultdict(list)
for r in rows:
    dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
pmc = json.load(open(pmc_json)) if os.path.exists(pmc_json) else {}
line = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
nh, npr_it = line["helm_iters_per_step"], line["pres_iters_per_step"]
from nekstab_amd.settings import PRODUCTION
geom = line.get("geometry") or dict(nel=1996, lx1=8, ndim=2, nvert=2033, coarse_lda=2048, patch_stride=100, nproj=PRODUCTION["nproj"])
if "geometry" not in line:                     # records of earlier rounds: E and lx1 from the workload string
    import re
    m = re.search(r"E=(\d+), lx1=(\d+)", line.get("config", {}).get("workload", ""))
    if m:
        nel_, lx1_ = int(m.group(1)), int(m.group(2))
        nv_ = int(round(nel_ * 2033.0 / 1996.0))
        geom = dict(nel=nel_, lx1=lx1_, ndim=2, nvert=nv_, coarse_lda=((nv_ + 255) // 256) * 256, patch_stride=(((lx1_ + 2) ** 2 + 3) // 4) * 4, nproj=PRODUCTION["nproj"])
geom = {k: v for k, v in geom.items() if k != "zero_arrays" or geom.get("ndim") == 3}
NX = geom["lx1"]
per = roofline.per_step_bytes(helm_iters=1.0, pres_iters=1.0, **geom)        # per iteration entries with counts = 1
perj = roofline.per_step_bytes(helm_iters=nh, pres_iters=npr_it, **geom)
P = geom["nel"] * NX ** 2
alg = {
    "k_convect<%d>" % NX: perj["K1 convect"], "k_rhs<%d>" % NX: perj["K2 rhs"], "k_helm<%d>" % NX: 148.0 * 2 * P, "k_pres_rhs<%d>" % NX: perj["K4 pres_rhs"],
    "k_coarse": perj["K6 coarse (x n_pres)"] / npr_it, "k_coarse_big": perj["K6 coarse (x n_pres)"] / npr_it, "k_schwarz<%d>" % NX: perj["K6 schwarz (x n_pres)"] / npr_it,
    "k_divgs<%d>" % NX: perj["K7 divgs (x n_pres)"] / npr_it, "k_gmres_update<%d>" % NX: perj["K7 gmres_update (x n_pres)"] / (npr_it + 1),
    # merged bookkeeping + coarse solve: the bytes of both (the restriction history is nvert doubles per basis vector: negligible)
    ("k_update_coarse<%d>" % (3 * ((geom["coarse_lda"] // 256 + 2) // 3))): perj["K6 coarse (x n_pres)"] / npr_it + perj["K7 gmres_update (x n_pres)"] / (npr_it + 1),
    # round 6: the two launches of a merged iteration
    ("k_schwarz_uc<%d, %d>" % (NX, 3 * ((geom["coarse_lda"] // 256 + 2) // 3))): (perj["K6 coarse (x n_pres)"] + perj["K6 schwarz (x n_pres)"]) / npr_it,
    "k_divgs_t<%d, false>" % NX: perj["K7 divgs (x n_pres)"] / npr_it, "k_divgs_t<%d, true>" % NX: perj["K7 divgs (x n_pres)"] / npr_it,
    "k_proj_apply_e<%d>" % NX: perj["projection apply/update"] * 0.4,
    "k_pres_update<%d>" % NX: perj["K10 pres_update"], "k_vel_update_proj<%d>" % NX: perj["K10 vel_update(+proj)"],
    "k_proj_apply": perj["projection apply/update"] * 0.4, "k_proj_update": perj["projection apply/update"] * 0.6,
}
tot = sum(sum(v) for v in dur.values())
print("| kernel | launches | p10 us | p50 us | p90 us | % of kernel time | algorithmic MB / launch | TB/s at p90 | frac of 8 TB/s | PMC MB / launch (2 x FETCH + WRITE, p90) |")
print("|---|---|---|---|---|---|---|---|---|---|")
traffic = {}
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    if len(v) < 100:
        continue
    short = k.replace("void ", "").replace("nsk::k2::", "").replace("nsk::", "")
    a = alg.get(short)
    rec = pmc.get(k)
    pm = (2.0 * rec["fetch_kb_p90"] + rec["write_kb_p90"]) * 1024.0 if rec else None
    if pm is not None:
        traffic[short] = {"bytes_per_launch": pm, "fetch_kb_p90": rec["fetch_kb_p90"], "write_kb_p90": rec["write_kb_p90"]}
    p10, p50, p90 = np.percentile(v, [10, 50, 90])
    print("| %s | %d | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s |" % (
        short, len(v), p10, p50, p90, 100 * v.sum() / tot, "%.2f" % (a / 1e6) if a else "-",
        "%.2f" % (a / p90 / 1e6) if a else "-", "%.2f" % (a / p90 / 1e6 / 8.0) if a else "-", "%.2f" % (pm / 1e6) if pm is not None else "-"))
stamp = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip.so.srchash")
json.dump({"srchash": open(stamp).read().strip() if os.path.exists(stamp) else None, "kernels": traffic,
           "source": "scripts/profile_r0N.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches), p90 over launches"},
          open(os.path.join(os.path.dirname(pmc_json), (os.path.basename(pmc_json).replace("_pmc_fetch_write_per_kernel.json", "") if "_pmc_fetch_write_per_kernel" in pmc_json else os.path.basename(pmc_json).split("_")[0]) + "_pmc_traffic.json"), "w"), indent=1)
