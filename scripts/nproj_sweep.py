"""cfg 2: matvec time and pressure iterations per step vs size of the pressure projection space."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
qx, qy = seed.add_noise(c)
for nproj in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 24, 32]:
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=nproj, schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
    v0, v1 = h.alloc(2)
    h.upload(v0, qx, qy, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0))
    ts = []
    for k in range(6):
        t0 = time.time(); h.matvec(v1, v0, 0); h.norm(v1); ts.append(time.time() - t0)
        st = h.stats()
        h.copy(v0, v1); h.scal(v0, 1.0 / h.norm(v0))
    print("nproj %2d: matvec %.1f ms (min of last 3), helm/step %.2f pres/step %.2f" % (nproj, 1e3 * min(ts[-3:]), st["helm_iters"] / h.nsteps, st["pres_iters"] / h.nsteps), flush=True)
    h.close()
