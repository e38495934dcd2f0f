"""Is the reference's committed lid-driven-cavity base flow (examples/lid_driven/BF_cav0.f00001) a fixed point of
the device's nonlinear map?  Scans the Reynolds number (the committed .par was edited after the file was written).
Needs /root/reference (run in the build container to make the fixture) or the fixture tests/golden/cavity_case.npz."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, nekio
from nekstab_amd.capi import NekStabHip
from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
fix = os.path.join(ROOT, "tests/golden/cavity_case.npz")
z = np.load(fix)
m = nekio.Re2Mesh(2, z["xc"].shape[0], z["xc"], z["yc"], None, [], [(int(a), int(b), np.zeros(5), str(c)) for (a, b), c in zip(z["bc_ef"], z["bc_code"])])
J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
for re in [float(a) for a in sys.argv[1:]] or [3600.0]:
    case = mesh.build_case_2d(m, z["vlex"].astype(np.int64), z["bf_u"].astype(np.float64), 6, re=re, endtime=1.0, spng_str=0.0)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=0, max_helm_iter=150, max_pres_iter=48)
    q, f = h.alloc(2)
    h.upload(q, case.ub[0], case.ub[1], J @ z["bf_p"].astype(np.float64) @ J.T)
    h.nonlinear_map(f, q, subtract_q=True)
    print("Re", re, "nsteps", h.nsteps, "|Phi(q)-q|^2 =", h.norm(f) ** 2, "|q|^2 =", h.norm(q) ** 2, h.stats()["unconverged"], flush=True)
    h.close()
