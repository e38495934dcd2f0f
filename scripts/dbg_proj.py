"""per-step projection diagnostics (NSK_DEBUG=1 NSK_USE_GRAPH=0): |g| before and |g'| after the projection onto the stored pressure solutions"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from nekstab_amd.capi import NekStabHip
from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
G = os.path.join(ROOT, "tests/golden")
which, nproj, tp = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
if which == "bfs":
    case = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 6, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
    tg = np.load(os.path.join(G, "backstep_tg.npz"))
    pu = tg["pRe_u"].astype(float); q0 = (pu[0], pu[1], J @ tg["pRe_p"].astype(float) @ J.T)
else:
    case = mesh.load_case_npz(os.path.join(G, "cylinder_case.npz"), 6, adjoint=(which == "cyla"))
    m = np.load(os.path.join(G, "cylinder_modes.npz"))
    u = m["dRe_u"].astype(float); q0 = (u[0], u[1], J @ m["dRe_p"].astype(float) @ J.T)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=tp, tol_relative=1, nproj=nproj, max_helm_iter=150, max_pres_iter=48)
h.set_option('proj_restart', int(os.environ.get('PROJ_RESTART', '1')))
q, f = h.alloc(2)
h.upload(q, *q0)
h.set_nsteps(int(os.environ.get('NST', '24')))
try:
    h.matvec(f, q, 1 if which == "cyla" else 0)
except Exception as e:
    print("ERR", e)
print(h.stats())
