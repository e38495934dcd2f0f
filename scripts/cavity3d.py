"""Lid-driven cavity (reference example, Re=3600): committed 2-D base flow extruded in z must be a fixed point of the
hexahedral nonlinear map; Newton-Krylov on the hexahedral context recovers it from a perturbed start."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d, nekio, newton
from nekstab_amd.capi import NekStabHip
from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
z = np.load(os.path.join(ROOT, "tests/golden/cavity_case.npz"))
m = nekio.Re2Mesh(2, z["xc"].shape[0], z["xc"], z["yc"], None, [], [(int(a), int(b), np.zeros(5), str(c)) for (a, b), c in zip(z["bc_ef"], z["bc_code"])])
J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
c2 = mesh.build_case_2d(m, z["vlex"].astype(np.int64), z["bf_u"].astype(np.float64), 6, re=3600.0, endtime=1.0, spng_str=0.0)
p2 = J @ z["bf_p"].astype(np.float64) @ J.T
nz = 2
c3 = mesh3d.extrude_case(c2, nz, 0.4, periodic=True)
t0 = time.time()
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=0, max_helm_iter=150, max_pres_iter=48)
print("init %.1fs nsteps %d" % (time.time() - t0, h.nsteps), flush=True)
q, f = h.alloc(2)
q3 = [mesh3d.extrude_field(c2.ub[0], nz), mesh3d.extrude_field(c2.ub[1], nz), np.zeros(c3.x.shape), mesh3d.extrude_pressure(p2, nz)]
h.upload3(q, *q3)
t0 = time.time(); h.nonlinear_map(f, q, subtract_q=True); t1 = time.time() - t0
print("3-D: |Phi(q)-q|^2 = %.3e (x 1/lz = %.3e)  |q|^2 = %.4f  %.2fs/map" % (h.norm(f) ** 2, h.norm(f) ** 2 / 0.4, h.norm(q) ** 2, t1), h.stats()["unconverged"], flush=True)
if len(sys.argv) > 1:
    # Newton from a perturbed start (interior scaled down, a little spanwise velocity)
    rng = np.random.default_rng(0)
    s = 1.0 - 0.05 * c3.mask
    h.upload3(q, q3[0] * s, q3[1] * s, 1e-3 * np.sin(2 * np.pi * c3.z / 0.4) * c3.mask * c3.x, q3[3])
    log = lambda *a: print(a, flush=True)
    t0 = time.time()
    it, hist = newton.newton_krylov(h, q, k_dim=40, tol=1e-10, maxiter_newton=8, log=log)
    out = h.download3(q)
    print("newton its", it, "hist", hist, "%.1fs" % (time.time() - t0))
    print("distance to the committed base flow: u %.2e v %.2e w %.2e" % tuple(np.abs(out[k] - q3[k]).max() for k in range(3)))
