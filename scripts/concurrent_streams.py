"""How much of the GPU does one latency-bound Arnoldi stream use?  R independent contexts (own HIP
stream, own state) driven by R host threads on one GPU; aggregate matvecs/s."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 6
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
qx, qy = seed.add_noise(case)
ctxs = []
for r in range(R):
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8,
                   max_helm_iter=100, max_pres_iter=48)
    a, b = h.alloc(2)
    h.upload(a, qx * (1 + 0.1 * r), qy, np.zeros(h.npres)); h.scal(a, 1.0 / h.norm(a))
    for _ in range(3):
        h.matvec(b, a, 0); h.copy(a, b); h.scal(a, 1.0 / h.norm(a))       # warm: budgets, graphs
    ctxs.append((h, a, b))
def work(h, a, b):
    for _ in range(nm):
        h.matvec(b, a, 0); h.copy(a, b); h.scal(a, 1.0 / h.norm(a))
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=c) for c in ctxs]
[t.start() for t in ths]; [t.join() for t in ths]
dt = time.perf_counter() - t0
print("R=%d streams: %d matvecs in %.3fs -> %.2f matvecs/s aggregate (%.1f ms per matvec per stream)" % (R, R * nm, dt, R * nm / dt, 1e3 * dt / nm))
