#!/usr/bin/env python3
"""Per-launch times of the launch-per-iteration CG kernel and of the persistent velocity solve (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), lx1)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-10, tol_pres=1e-1, tol_relative=1, schwarz_layers=2, max_helm_iter=100, max_pres_iter=48, nproj=8)
h.set_option("min_pres_iter", 2)
qx, qy = seed.add_noise(case)
q, f = h.alloc(2)
h.upload(q, qx, qy, np.zeros(h.npres)); h.scal(q, 1.0 / h.norm(q))
import time
for fused in (1, 0, 1, 0):
    h.set_option("fused", fused)
    h.matvec(f, q, 0); h.matvec(f, q, 0)
    t0 = time.perf_counter(); h.matvec(f, q, 0); h.norm(f); t1 = time.perf_counter() - t0
    print("fused", fused, "matvec %.1f ms  %.1f us/step" % (1e3 * t1, 1e6 * t1 / h.nsteps), h.stats())
h.set_option("fused", 1)
a = h.bench_kernel("helm", 200); b = h.bench_kernel("helm_fused", 200); c = h.bench_kernel("helm_fused0", 200)
print("k_helm per launch %.2f us; fused with 8 iterations %.2f us, with 0 iterations %.2f us => %.2f us per iteration" % (a["avg_us"], b["avg_us"], c["avg_us"], (b["avg_us"] - c["avg_us"]) / 8))

for dbg in (1, 3):
    h.set_option("dbg", dbg)
    b = h.bench_kernel("helm_fused", 200); c = h.bench_kernel("helm_fused0", 200)
    print("ablation dbg=%d (1: no grid barrier, 2: no partial-sum loads): 8 iterations %.2f us, 0 iterations %.2f us => %.2f us per iteration" % (dbg, b["avg_us"], c["avg_us"], (b["avg_us"] - c["avg_us"]) / 8))
h.set_option("dbg", 0)
