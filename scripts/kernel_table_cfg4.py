"""Per-kernel table of config 4 (E = 50 100 hexahedra, lx1 = 8): HIP-event timings of every hot kernel launched back to back at a
known basis index (scripts/kernels3d_bench.py), the algorithmic bytes of that launch (nekstab_amd/roofline.py: every distinct
array once), the HBM-side bytes of the SAME launches from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), and
the fractions of 8 TB/s by either byte count; below it the kernel's share of a time step from the steady-state trace.
Usage: kernel_table_cfg4.py <kernels3d_bench output> <pmc_summary.json> [<trace_dir>]"""
import collections, csv, glob, json, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import roofline
txt, pmc_json = sys.argv[1:3]
trace_dir = sys.argv[3] if len(sys.argv) > 3 else None
t = {}
for l in open(txt):
    m = re.match(r"^(\w+)\s+([\d.]+) us", l)
    if m:
        t[m.group(1)] = float(m.group(2))
    m = re.match(r"^E = (\d+), P = ([\d.]+) M, P2 = ([\d.]+) M", l)
    if m:
        nel = int(m.group(1))
    m = re.match(r"^zero_arrays = 0x([0-9a-f]+)", l)
    if m:
        za = int(m.group(1), 16)
N, M = 8, 6
za = globals().get("za", 0)        # arrays that vanish on every node: not loaded, not counted (nekstab_amd/roofline.py)
P, P2 = nel * N ** 3, nel * M ** 3
nvert = 53670
one = roofline.per_step_bytes(nel=nel, lx1=N, ndim=3, nvert=nvert, nproj=0, helm_iters=0.0, pres_iters=1.0, pres_jsum=0.0, coarse_bytes=0.0, zero_arrays=za)
rule, distinct = roofline.helm_launch_bytes(nel=nel, lx1=N, ndim=3, zero_arrays=za)
stepb = roofline.per_step_bytes(nel=nel, lx1=N, ndim=3, nvert=nvert, nproj=32, helm_iters=0.0, pres_iters=0.0, pres_jsum=0.0, coarse_bytes=0.0, zero_arrays=za)
alg = {"helm": (distinct, "k3::k_helm<8>", "all arrays of the three components once (SURVEY rule, 172 B/pt and component: %.2f GB)" % (rule / 1e9)),
       "divgs": (one["K7 divgs (x n_pres)"], "k3::k_divgs<8>", "E apply without dots"),
       "schwarz": (one["K6 schwarz (x n_pres)"] - nel * 8 * 12.0, "k3::k_schwarz_w16<8>", "fast-diagonalisation Schwarz + D^T, one wavefront per element, sixteen per CU"),
       "schwarz_wg": (one["K6 schwarz (x n_pres)"] - nel * 8 * 12.0, "k3::k_schwarz<8>", "the same as one workgroup per element (round 3's form; option eapply_pipe = 0)"),
       "pres_rhs": (stepb["K4 pres_rhs"], "k3::k_pres_rhs<8>", "nproj = 32"), "rhs": (stepb["K2 rhs"], "k3::k_rhs<8>", ""),
       "convect_mfma": (stepb["K1 convect"], "k3::k_convect_mfma8", "")}
for j in (3, 8, 16, 24, 32):
    alg["gs_dots%d" % j] = (8.0 * P2 * (j + 2), None, "V_0..%d and w" % j)
    alg["gs_lag%d" % j] = (8.0 * P2 * (j + 3) + 64.0 * nel, None, "V_0..%d, w in; w' out (no pending correction in this timing)" % j)
gsname = {"gs_dots8": "k_gs_dots<8, 16, 2>", "gs_lag8": "k_gs_lag<8, 8, 4>", "gs_dots24": "k_gs_dots<8, 32, 1>", "gs_lag24": "k_gs_lag<8, 32, 1>"}
pmc = json.load(open(pmc_json)) if os.path.exists(pmc_json) else {}
def pmc_of(key):
    for k, v in pmc.items():
        if key and key.replace("k3::", "") in k:
            return (2.0 * v["fetch_kb_p50"] + v["write_kb_p50"]) * 1024.0
    return None
print("| kernel (launch) | HIP-event us | algorithmic GB / launch | TB/s | frac of 8 TB/s | counter GB / launch (2 x FETCH + WRITE) | frac by counter bytes | note |")
print("|---|---|---|---|---|---|---|---|")
out = {}
for kn in ("helm", "divgs", "schwarz", "schwarz_wg", "gs_dots3", "gs_lag3", "gs_dots8", "gs_lag8", "gs_dots16", "gs_lag16", "gs_dots24", "gs_lag24", "gs_dots32", "gs_lag32", "pres_rhs", "rhs", "convect_mfma"):
    if kn not in t or kn not in alg:
        continue
    a, key, note = alg[kn]
    pm = pmc_of(key or gsname.get(kn))
    us = t[kn]
    print("| %s | %.1f | %.3f | %.2f | %.2f | %s | %s | %s |" % (kn, us, a / 1e9, a / us / 1e6, a / us / 1e6 / 8.0,
          "%.3f" % (pm / 1e9) if pm else "-", "%.2f" % (pm / us / 1e6 / 8.0) if pm else "-", note))
    if pm:
        out["k3::" + kn] = {"bytes_per_launch": pm}
json.dump(out, open(os.path.join(os.path.dirname(pmc_json), "r04_cfg4_pmc_traffic_part.json"), "w"), indent=1)
if trace_dir and glob.glob(trace_dir + '/*/*kernel_trace.csv'):
    rows = list(csv.DictReader(open(glob.glob(trace_dir + '/*/*kernel_trace.csv')[0])))
    t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
    rows = [r for r in rows if int(r['Start_Timestamp']) >= t1 - 0.22 * (t1 - t0)]
    d = collections.defaultdict(float); n = collections.Counter()
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace("void ", "").replace("nsk::", "")
        d[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; n[k] += 1
    nsteps = max(n.get("k3::k_convect_mfma8", 1), 1)
    tot = sum(d.values())
    print("\nShare of a time step (steady-state trace: the last of four 40-step maps, %d steps in the window, %.1f ms of kernel time per step):\n" % (nsteps, tot / nsteps / 1e3))
    print("| kernel | launches per step | ms per step | % |")
    print("|---|---|---|---|")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
        if v / tot > 0.003:
            print("| %s | %.1f | %.2f | %.1f |" % (k, n[k] / nsteps, v / nsteps / 1e3, 100 * v / tot))
