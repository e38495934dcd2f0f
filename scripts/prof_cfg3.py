"""Profile driver for BASELINE config 3 on one GPU (E=7984, lx1=12): a short map, eager launches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c0 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
case = mesh.refine_case_2x2(c0)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
u0 = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1)
c1 = mesh.refine_case_2x2(mesh.Case(**{**c0.__dict__, "ub": u0}))          # the mode, split like the mesh
v0, v1 = h.alloc(2)
h.upload(v0, c1.ub[0] * case.mask, c1.ub[1] * case.mask, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0))
h.set_nsteps(nst)
for k in range(2):
    t0 = time.time(); h.matvec(v1, v0, 0); h.norm(v1); dt = time.time() - t0
    st = h.stats()
    print("map %d: %.2f ms/step helm/step %.1f pres/step %.1f" % (k, 1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst), flush=True)
