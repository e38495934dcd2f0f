import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/cylinder_case.npz"), 8)
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
qx, qy = seed.add_noise(c)
v0, v1 = h.alloc(2)
h.upload(v0, qx, qy, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0))
for k in range(4):
    t0 = time.time(); h.matvec(v1, v0, 0); h.norm(v1); dt = time.time() - t0
    h.copy(v0, v1); h.scal(v0, 1.0 / h.norm(v0))
    print("matvec %.1f ms" % (1e3 * dt), flush=True)
print(os.environ.get("NSK_LIB", "default"), h.bench_kernel("helm", 400))
