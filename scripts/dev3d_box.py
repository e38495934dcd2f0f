"""Pressure-solve iteration counts on small boxes (development)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh3d
from nekstab_amd.capi import NekStabHip
for warp in (0.0, 0.06):
    for n in (6, 8):
        c = mesh3d.box_case_3d(3, 3, 3, n, lengths=(1.0, 1.5, 0.9), outflow_xmax=True, re=10., endtime=0.01,
                               ub_func=lambda x, y, z: np.stack([1 + 0 * x, 0 * x, 0 * x]), warp=warp)
        h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-12, tol_pres=1e-8, tol_relative=1, max_helm_iter=200, max_pres_iter=48)
        g = np.random.default_rng(0).standard_normal((c.nel,) + (n - 2,) * 3)
        x, it = h.t_pres_solve(g)
        r = h.t_eapply(x) - g
        print("warp", warp, "lx1", n, "iters", it, "true rel res", np.linalg.norm(r) / np.linalg.norm(g), flush=True)
        h.close()
