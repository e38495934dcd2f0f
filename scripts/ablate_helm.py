import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from nekstab_amd.capi import NekStabHip
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-300, tol_pres=1e-3, tol_relative=0, max_helm_iter=100, max_pres_iter=48)
rng = np.random.default_rng(0)
rx = rng.standard_normal(case.x.shape); ry = rng.standard_normal(case.x.shape)
for dbg in (0, 1, 2, 4, 8, 15, 0):
    h.set_option("dbg", dbg)
    h.t_helm_solve(rx, ry, 3)
    t0 = time.time()
    for _ in range(5): h.t_helm_solve(rx, ry, 3)
    dt = (time.time() - t0) / 5
    print("dbg %2d: %.1f us per k_helm launch (100 per solve, eager, incl. host)" % (dbg, dt / 100 * 1e6))
