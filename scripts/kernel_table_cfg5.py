"""Per-kernel table of config 5 (E = 99 452 hexahedra, lx1 = 10): HIP-event timings of the hot kernels (scripts/prof_cfg5.py, KERNELS=...),
their algorithmic bytes (nekstab_amd/roofline.py) and the HBM-side bytes of the same launches from the PMC passes (2 x FETCH_SIZE +
WRITE_SIZE).  Writes <dir>/r06_cfg5_pmc_traffic_part.json.  Usage: kernel_table_cfg5.py <prof_cfg5 output> <pmc_summary.json>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import roofline
txt, pmc_json = sys.argv[1:3]
t, za, nel, steps = {}, 0, 99452, []
for l in open(txt):
    m = re.match(r"^(\w+)\s+([\d.]+) us", l)
    if m:
        t[m.group(1)] = float(m.group(2))
    m = re.match(r"^E (\d+) set-up", l)
    if m:
        nel = int(m.group(1))
    m = re.match(r"^zero_arrays = 0x([0-9a-f]+)", l)
    if m:
        za = int(m.group(1), 16)
    m = re.match(r"^([\d.]+) ms per step \(([\d.]+) Helmholtz \+ ([\d.]+) pressure", l)
    if m:
        steps.append((float(m.group(1)), float(m.group(2)), float(m.group(3))))
N = 10
one = roofline.per_step_bytes(nel=nel, lx1=N, ndim=3, nvert=0, nproj=0, helm_iters=0.0, pres_iters=1.0, pres_jsum=0.0, coarse_bytes=0.0, zero_arrays=za)
rule, distinct = roofline.helm_launch_bytes(nel=nel, lx1=N, ndim=3, zero_arrays=za)
conv = one["K1 convect"]                                   # (the base-flow constants that vanish on the whole mesh are not counted: zero_arrays)
zm = bin(za & 0x1ff).count("1")                             # metric terms that vanish on the whole mesh
conv_nl = 8.0 * (nel * N ** 3 * 8 + nel * (3 * N // 2) ** 3 * (9 - zm))      # u (3) in, bf (3) out, sponge, mass on the GLL mesh; the nine metric terms on the dealiasing mesh
# (bench name, kernel-name key of the PMC table, algorithmic bytes, note); the first of a group is the form the context runs
rows = [("helm", "k_helm_p<10>", distinct, "one CG iteration of the three components, resident workgroups + LDS-DMA prefetch (round 6); all arrays once (SURVEY rule: %.2f GB)" % (rule / 1e9)),
        ("helm_wg", "k_helm<10>", distinct, "... one workgroup per element (rounds 3-5)"),
        ("divgs", "k_divgs_c3<10>", one["K7 divgs (x n_pres)"], "E apply, components side by side: two workgroups per CU (default at lx1 = 10 since round 6)"),
        ("divgs_wg", "k_divgs<10>", one["K7 divgs (x n_pres)"], "... one workgroup per CU (rounds 3-5)"),
        ("schwarz", "k_schwarz_q<10>", one["K6 schwarz (x n_pres)"] - nel * 8 * 12.0, "Schwarz + D^T, four wavefronts per element, six elements per CU (round 6)"),
        ("schwarz_p", "k_schwarz_p<10>", one["K6 schwarz (x n_pres)"] - nel * 8 * 12.0, "... resident 1024-thread workgroups"),
        ("schwarz_wg", "k_schwarz<10>", one["K6 schwarz (x n_pres)"] - nel * 8 * 12.0, "... one 1024-thread workgroup per element (rounds 3-5)"),
        ("convect_mfma", "k_convect_mfma<10>", conv, "dealiased convection on the matrix cores (round 6)"),
        ("convect", "k_convect<10>", conv, "... thread per node, constants loaded inside the point loop (rounds 2-5)"),
        ("convect_mfma_nl", "k_convect_mfma_nl<10>", conv_nl, "the FULL equations' convection term (mode 2: Newton-Krylov) on the matrix cores (round 6)"),
        ("convect_nl", "k_convect<10>", conv_nl, "... thread per node (mode 2 of k_convect<10>)")]
pmc = json.load(open(pmc_json)) if os.path.exists(pmc_json) else {}
def pmc_of(key):
    for k, v in pmc.items():
        if "k3::" + key in k:
            return (2.0 * v["fetch_kb_p50"] + v["write_kb_p50"]) * 1024.0
    return None
for ms, hi, pi in steps[-1:]:
    print("config 5 at full size: %.0f ms per time step (%.1f Helmholtz + %.1f pressure iterations per step); zero_arrays 0x%x\n" % (ms, hi, pi, za))
print("| kernel (launch) | HIP-event us | algorithmic GB / launch | TB/s | frac of 8 TB/s | counter GB / launch (2 x FETCH + WRITE) | frac by counter bytes | note |")
print("|---|---|---|---|---|---|---|---|")
out = {}
for kn, key, a, note in rows:
    if kn not in t:
        continue
    pm = None if kn in ("convect_nl", "convect") else pmc_of(key)      # (k_convect<10> runs in two modes under the counters: no per-mode figure)
    us = t[kn]
    print("| k3::%s | %.1f | %.3f | %.2f | %.2f | %s | %s | %s |" % (key, us, a / 1e9, a / us / 1e6, a / us / 1e6 / 8.0, "%.3f" % (pm / 1e9) if pm else "-", "%.2f" % (pm / us / 1e6 / 8.0) if pm else "-", note))
    if pm and kn in ("helm", "divgs", "schwarz", "convect_mfma"):
        out["k3::%s<10>" % kn] = {"bytes_per_launch": pm, "kernel": key}
json.dump(out, open(os.path.join(os.path.dirname(pmc_json), "r06_cfg5_pmc_traffic_part.json"), "w"), indent=1)
