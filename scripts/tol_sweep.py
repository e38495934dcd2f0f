"""cfg 2: accuracy of one full matvec (183 steps) and its cost as a function of the pressure / Helmholtz tolerances,
against a tightly converged run (relative bm1-L2 difference of the velocity)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
u0 = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), 8) * c.mask
qx, qy = seed.add_noise(c)
def run(th, tp, q, em=None, mp=0, cap=0):
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=th, tol_pres=tp, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
    if em is not None:
        h.set_option('early_pres_mul', em)
    if mp:
        h.set_option('min_pres_iter', mp)
    if cap:
        h.set_option('pres_cap', cap)
    v0, v1 = h.alloc(2)
    h.upload(v0, q[0], q[1], np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0))
    h.matvec(v1, v0, 0)
    ts = []
    for k in range(3):
        t0 = time.time(); h.matvec(v1, v0, 0); h.norm(v1); ts.append(time.time() - t0)
    out = h.download(v1); st = h.stats(); w = np.array(h._keep["x"]) * 0
    h.close()
    return out, min(ts), st
def krylov_vec(n):
    from nekstab_amd import krylov
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-12, tol_pres=1e-5, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=150, max_pres_iter=48)
    Q = h.alloc(n + 2); H = np.zeros((n + 2, n + 1))
    h.upload(Q[0], qx, qy, np.zeros(h.npres)); h.scal(Q[0], 1.0 / h.norm(Q[0]))
    krylov.arnoldi_factorization(h, Q, H, 1, n, 0)
    out = h.download(Q[n]); h.close()
    return out[0], out[1]
for name, q in (("mode", u0), ("K12", krylov_vec(12)), ("noise", (qx, qy))):
    ref, _, _ = run(1e-13, 1e-7, q, 1.0)
    for th, tp, em, mp, cap in ((1e-9, 3e-1, 1e-2, 2, 0), (1e-9, 3e-1, 1e-2, 2, 4), (1e-9, 3e-1, 1e-2, 2, 3), (1e-10, 3e-1, 1e-2, 2, 4), (1e-9, 1.0, 1e-2, 2, 3)):
        out, t, st = run(th, tp, q, em, mp, cap)
        err = np.sqrt(sum(np.sum((a - b) ** 2) for a, b in zip(out[:2], ref[:2])) / sum(np.sum(b ** 2) for b in ref[:2]))
        print("%-5s cap %d min_pres %d early %.0e tol_helm %.0e tol_pres %.0e: rel diff %.2e  %.1f ms  helm/step %.2f pres/step %.2f" % (name, cap, mp, em, th, tp, err, 1e3 * t, st["helm_iters"] / st["steps"], st["pres_iters"] / st["steps"]), flush=True)
