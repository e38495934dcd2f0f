"""lx1 = 6 cylinder (no projection space, as the test fixture): leading Ritz value of a 170-vector Krylov-Schur run for a
few production-tolerance candidates (reference Spectre_Hd.dat row 1: 0.7387113 + 0.6972442i)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 6)
for nproj in (8,):
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-13, tol_pres=1e-13, tol_relative=0, schwarz_layers=2, max_helm_iter=120, max_pres_iter=48, nproj=nproj)
    for th, tp, mp, cap in ((1e-10, 2e-1, 0, 0), (1e-9, 3e-1, 2, 0), (1e-9, 3e-1, 2, 4), (1e-9, 3e-1, 2, 3), (1e-9, 2e-1, 2, 4)):
        h.set_tolerances(th, tp, 1); h.set_option("min_pres_iter", mp); h.set_option("pres_cap", cap)
        qx, qy = seed.add_noise(c)
        v = h.alloc(1)[0]
        h.upload(v, qx, qy, np.zeros(h.npres))
        import time; t0 = time.time()
        try:
            res = krylov.krylov_schur(h, v, 170, schur_tgt=0)
            mu = res.vals[0] if res.vals[0].imag > 0 else res.vals[1]
            print("nproj %d tol %.0e/%.0e min %d cap %d: mu = %.7f %+.7fi  |mu-ref| %.1e  %.1fs" % (nproj, th, tp, mp, cap, mu.real, mu.imag, abs(mu - complex(0.7387113, 0.6972442)), time.time() - t0), flush=True)
            h.free(res.Q + [v])
        except Exception as ex:
            print("nproj %d tol %.0e/%.0e min %d: FAILED %s" % (nproj, th, tp, mp, ex), flush=True)
    h.close()
