#!/usr/bin/env python3
"""Back-to-back kernel timings of the quadrilateral time step on config 2 (nsk_bench_kernel, HIP events, full-work launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
h = production_context(case)
qx, qy = seed.add_noise(case)
Q = h.alloc(7); H = np.zeros((7, 6))
h.upload(Q[0], qx, qy, np.zeros(h.npres)); h.scal(Q[0], 1.0 / h.norm(Q[0]))
try:
    krylov.arnoldi_factorization(h, Q, H, 1, 6, 0)          # (only to leave realistic data in the solver arrays)
except Exception as exc:                                   # noqa: BLE001 (experimental builds may not converge)
    print("note:", exc)
for kn in ("helm", "convect", "rhs", "pres_rhs", "proj_apply", "gmres_update", "update_coarse0", "update_coarse3", "update_coarse8", "schwarz", "divgs2", "divgs", "coarse", "pres_chain_merged", "pres_chain", "pres_update", "vel_update_proj", "proj_update"):
    try:
        print("%-20s %8.2f us" % (kn, h.bench_kernel(kn, 200)["avg_us"]), flush=True)
    except Exception as e:
        print(kn, "failed", e)
h.close()
