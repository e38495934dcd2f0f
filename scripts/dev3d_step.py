"""z-extruded backward-facing step (config 4's geometry): hexahedral direct map vs the committed optimal response."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
G = os.path.join(ROOT, "tests/golden")
case = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 6, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
tg = np.load(os.path.join(G, "backstep_tg.npz"))
J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lz = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c3 = mesh3d.extrude_case(case, nz, lz, periodic=True)
pu = tg["pRe_u"].astype(float); ore = tg["ore_u"].astype(float)
for tp in (1e-3, 1e-5, 1e-7):
    h3 = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-11, tol_pres=tp, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=48)
    q, f = h3.alloc(2)
    h3.upload3(q, mesh3d.extrude_field(pu[0], nz), mesh3d.extrude_field(pu[1], nz), np.zeros(c3.x.shape), mesh3d.extrude_pressure(J @ tg["pRe_p"].astype(float) @ J.T, nz))
    t0 = time.time()
    try:
        h3.matvec(f, q, 0)
    except Exception as e:
        print("ERR", e)
    dt = time.time() - t0
    out = h3.download3(f); st = h3.stats()
    sc = np.abs(ore).max(); e = slice(0, case.nel)
    print("nz", nz, "tolp", tp, "err u %.2e v %.2e w %.2e gain %.5f" % (np.abs(out[0][e, 2] - ore[0]).max() / sc, np.abs(out[1][e, 2] - ore[1]).max() / sc, np.abs(out[2]).max() / sc, h3.norm(f) ** 2 / lz),
          "helm/step %.1f pres/step %.1f maxp %d unconv %d  %.2fs" % (st["helm_iters"] / h3.nsteps, st["pres_iters"] / h3.nsteps, st["max_pres_iter"], st["unconverged"], dt), flush=True)
    h3.close()
