"""BASELINE config 3 on one GPU: cylinder mesh refined 2x2 (E=7984), lx1=12 (N=11): timing of the matvec."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 12
c0 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
case = mesh.refine_case_2x2(c0)
t0 = time.time()
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-10, tol_pres=2e-1, tol_relative=1, nproj=8,
               schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
print("E=%d lx1=%d points/field=%d init %.1fs dt=%g nsteps=%d" % (case.nel, lx1, h.nvel, time.time() - t0, h.dt, h.nsteps), flush=True)
qx, qy = seed.add_noise(case)
v0, v1 = h.alloc(2)
h.upload(v0, qx, qy, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0))
for k in range(3):
    t0 = time.time()
    try:
        h.matvec(v1, v0, 0)
    except Exception as e:
        print('ERR', e, h.stats()); break
    dt = time.time() - t0
    st = h.stats()
    print("matvec %d: %.3fs  (%.1f us/step)  helm/step %.1f pres/step %.1f budgets %d/%d" % (k, dt, 1e6 * dt / h.nsteps, st["helm_iters"] / h.nsteps, st["pres_iters"] / h.nsteps, st["budget_helm"], st["budget_pres"]), flush=True)
    h.copy(v0, v1); h.scal(v0, 1.0 / h.norm(v0))
print(h.bench_kernel("helm", 100))
