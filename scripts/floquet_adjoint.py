import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.capi import NekStabHip
from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
z = np.load(os.path.join(ROOT, "tests/golden/cylinder_upo.npz"))
T = float(z["period"])
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1, endtime=T)
u = mesh.interp_field_2d(z["u"], lx1); p1 = mesh.interp_field_2d(z["p"], lx1)
case.ub[:] = u
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-2, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=48)
J = interp_matrix(gauss_lobatto_legendre(lx1)[0], gauss_legendre(lx1 - 2)[0])
q0, qe = h.alloc(2)
h.upload(q0, u[0], u[1], J @ p1 @ J.T)
t0 = time.time(); h.set_orbit(q0, spng_str=1.7, end=qe); print("orbit: nsteps", h.nsteps, "time %.1fs" % (time.time() - t0))
qx, qy = seed.add_noise(case)
v0, v1 = h.alloc(2)
h.upload(v0, qx, qy, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0)); h.matvec(v1, v0, mode)
t0 = time.time(); res = krylov.krylov_schur(h, v1, 30, mode=mode, schur_tgt=0); print("arnoldi %.1fs" % (time.time() - t0))
for i in range(8): print(res.vals[i], res.residual[i])
