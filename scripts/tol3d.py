"""Hexahedral path (z-extruded cylinder, lx1 = 8): accuracy and cost of one 183-step map at the production tolerances
against a tightly converged run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
lx1, nz = 8, 2
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
c3 = mesh3d.extrude_case(c2, nz, 0.5 * nz, periodic=True)
u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1) * c2.mask
rng = np.random.default_rng(0)
w = 1e-3 * rng.standard_normal(c3.x.shape) * c3.mask
q = (mesh3d.extrude_field(u[0], nz), mesh3d.extrude_field(u[1], nz), w)
def run(th, tp, mp, cap=0):
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=th, tol_pres=tp, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=8)
    if mp: h.set_option("min_pres_iter", mp)
    if cap: h.set_option("pres_cap", cap)
    b0, b1 = h.alloc(2)
    h.upload3(b0, q[0], q[1], q[2], np.zeros(h.npres))
    ts = []
    for rep in range(3):
        t0 = time.time(); h.matvec(b1, b0, 0); ts.append(time.time() - t0)
    out = h.download3(b1); st = h.stats(); h.close()
    return out, min(ts), st
ref, _, _ = run(1e-12, 1e-3, 0)
for th, tp, mp, cap in ((1e-9, 3e-1, 2, 0), (1e-9, 3e-1, 2, 8), (1e-9, 3e-1, 2, 6), (1e-9, 3e-1, 3, 5), (1e-9, 2e-1, 2, 8)):
    out, t, st = run(th, tp, mp, cap)
    err = np.sqrt(sum(np.sum((a - b) ** 2) for a, b in zip(out[:3], ref[:3])) / sum(np.sum(b ** 2) for b in ref[:3]))
    print("tol_helm %.0e tol_pres %.0e min_pres %d cap %d: rel diff %.2e  %.2f ms/step  helm/step %.2f pres/step %.2f" % (th, tp, mp, cap, err, 1e3 * t / 183, st["helm_iters"] / st["steps"], st["pres_iters"] / st["steps"]), flush=True)
