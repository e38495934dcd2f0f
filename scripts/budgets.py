"""cfg 2: launched iteration budgets vs iterations actually needed (graph replay)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=100, max_pres_iter=48)
qx, qy = seed.add_noise(c)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
Q = h.alloc(n + 2)
h.upload(Q[0], qx, qy, np.zeros(h.npres)); h.scal(Q[0], 1.0 / h.norm(Q[0]))
H = np.zeros((n + 2, n + 1))
for m in range(1, n + 1):
    t0 = time.time()
    krylov.arnoldi_factorization(h, Q, H, m, m, 0)
    dt = time.time() - t0
    st = h.stats()
    if m % 4 == 0 or st["retries"]:
        print("m=%3d %.1f ms helm/step %.2f (max %d, budget %d) pres/step %.2f (max %d, budget %d) retries %d recaptures %d" % (
            m, 1e3 * dt, st["helm_iters"] / st["steps"], st["max_helm_iter"], st["budget_helm"], st["pres_iters"] / st["steps"], st["max_pres_iter"], st["budget_pres"], st["retries"], st["recaptures"]), flush=True)
