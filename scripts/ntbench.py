"""cfg 2: a few matvecs + the k_helm launch time for the library named by NSK_LIB (A/B runs on one box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
qx, qy = seed.add_noise(c)
v0, v1 = h.alloc(2)
h.upload(v0, qx, qy, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0))
ts = []
for k in range(8):
    t0 = time.time(); h.matvec(v1, v0, 0); h.norm(v1); ts.append(time.time() - t0)
    h.copy(v0, v1); h.scal(v0, 1.0 / h.norm(v0))
print(os.path.basename(os.environ.get("NSK_LIB", "default")), "matvec ms:", " ".join("%.1f" % (1e3 * t) for t in ts), h.bench_kernel("helm", 400)["avg_us"])
