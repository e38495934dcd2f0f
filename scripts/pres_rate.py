"""Convergence of the preconditioned GMRES pressure solve (cfg 2 mesh, test hook): iterations to reach a relative
residual, for a smooth right-hand side, noise, and the divergence of a noise velocity field."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-10, tol_pres=1e-1, tol_relative=1, nproj=0, schwarz_layers=layers, max_helm_iter=100, max_pres_iter=48)
rng = np.random.default_rng(0)
m = lx1 - 2
xs = h.t_opdiv(c.x * c.mask, 0 * c.x)           # some smooth field on the pressure mesh
qx, qy = seed.add_noise(c)
rhs = {"smooth": xs, "noise": rng.standard_normal(xs.shape), "div(noise velocity)": h.t_opdiv(qx, qy)}
for name, g in rhs.items():
    out = []
    for tol in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6):
        h.set_tolerances(1e-10, tol, 1)
        y, it = h.t_pres_solve(g)
        r = g - h.t_eapply(y)
        out.append("%g: %d it (true %.1e)" % (tol, it, np.linalg.norm(r) / np.linalg.norm(g)))
    print("%-22s" % name, " | ".join(out), flush=True)
