"""closed-domain (pressure null space) solves with a projection space: per-step diagnostics (NSK_DEBUG=1 NSK_USE_GRAPH=0)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from nekstab_amd.capi import NekStabHip
from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
G = os.path.join(ROOT, "tests/golden")
case = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 6, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
tg = np.load(os.path.join(G, "backstep_tg.npz"))
J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-5, tol_relative=1, nproj=int(sys.argv[1]) if len(sys.argv) > 1 else 8, max_helm_iter=150, max_pres_iter=48)
q, f = h.alloc(2)
pu = tg["pRe_u"].astype(float)
h.upload(q, pu[0], pu[1], J @ tg["pRe_p"].astype(float) @ J.T)
h.set_nsteps(24)
try:
    h.matvec(f, q, 0)
except Exception as e:
    print("ERR", e)
print(h.stats())
