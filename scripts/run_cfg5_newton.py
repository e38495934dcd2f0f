#!/usr/bin/env python3
"""BASELINE configs[4] at full size on one GPU as what it is meant to be -- a Newton-Krylov fixed-point computation followed by
a stability factorisation (core/newton_krylov.f:5-296, restated in nekstab_amd/newton.py): lid-driven cube, 46 x 46 x 47
hexahedra with cav.box's wall clustering, lx1 = 10 (99.5 M points per field, 2.79 GB per Krylov vector).  Two Newton iterations
with an 8-vector GMRES each from the crudest start there is (fluid at rest under the moving lid), then three Arnoldi steps of the
linearised operator about the last iterate.  The sampling period is cut to 24 time steps so that the run fits the GPU budget of
a development round (a production run uses O(1) convective times: hours on one GPU, the reference runs it on 8).

    python scripts/run_cfg5_newton.py [newton_iterations=2] [k_dim=8] [steps=24]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh3d, newton
from nekstab_amd.capi import NekStabHip
nnewt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kdim = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 24
n = [int(x) for x in os.environ.get("BOX", "46 46 47").split()]
stretch = lambda xi: 0.5 * (1.0 - np.cos(np.pi * xi))
lid = lambda x, y, z: np.stack([np.where(np.isclose(y, 1.0), (1.0 - (2 * x - 1.0) ** 2) ** 2 * (1.0 - (2 * z - 1.0) ** 2) ** 2, 0.0), 0.0 * x, 0.0 * x])
t0 = time.time()
c = mesh3d.box_case_3d(n[0], n[1], n[2], 10, lengths=(1.0, 1.0, 1.0), re=1000.0, endtime=0.05, stretch=stretch, ub_func=lid)
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-9, tol_pres=1e-3, tol_relative=1, max_helm_iter=400, max_pres_iter=192, nproj=8)
print("E %d, lx1 10: %d points per field, state %.2f GB; mesh + set-up %.0f s; dt %.3e, %d time steps per period (cut to %d)" % (c.nel, h.nvel, 8e-9 * h.nstate, time.time() - t0, h.dt, h.nsteps, nst), flush=True)


class Cut:
    """the backend with the sampling period cut to `nst` time steps after every re-linearisation"""
    def __init__(self, h): self.h = h
    def __getattr__(self, k): return getattr(self.h, k)
    def set_baseflow(self, q):
        self.h.set_baseflow(q)
        self.h.set_nsteps(min(self.h.nsteps, nst))


be = Cut(h)
tl = [time.time()]
def log(kind, i, r):
    now = time.time()
    s = h.stats()
    print("  %-8s %2d  residual^2 %.3e   %.0f s  (last map: %.1f Helmholtz + %.1f pressure iterations per step)" % (kind, i, r, now - tl[0], s["helm_iters"] / max(s["steps"], 1), s["pres_iters"] / max(s["steps"], 1)), flush=True)
    tl[0] = now
q = h.alloc(1)[0]
m = c.lx1 - 2
h.upload3(q, c.ub[0], c.ub[1], c.ub[2], np.zeros((c.nel, m, m, m)))
t0 = time.time()
# newton.newton_krylov with ONE GMRES cycle of k_dim vectors per Newton iteration (the reference restarts the cycle until its
# tolerance is met: a matter of GPU hours at this size, not of the algorithm)
fq, dq = h.alloc(2)
hist = []
for its in range(1, nnewt + 1):
    be.set_baseflow(q)
    h.nonlinear_map(fq, q, subtract_q=True)
    hist.append(h.norm(fq) ** 2)
    log("newton", its, hist[-1])
    newton.ts_gmres(be, fq, dq, kdim, 1e-14, maxiter=1, log=log)
    h.axpy(q, -1.0, dq)
h.free([fq, dq])
print("Newton: %d iterations in %.0f s, |Phi_T(q) - q|^2 at the start of each: %s" % (its, time.time() - t0, " ".join("%.3e" % r for r in hist)), flush=True)
# residual of the last iterate and the stability factorisation about it
f = h.alloc(1)[0]
be.set_baseflow(q)
h.nonlinear_map(f, q, subtract_q=True)
print("after the last update: |Phi_T(q) - q|^2 = %.3e" % (h.norm(f) ** 2), flush=True)
k = 3
Q = h.alloc(k + 1)
h.seed_noise(Q[0]) if hasattr(h, "seed_noise") else None
h.scal(Q[0], 1.0 / h.norm(Q[0]))
H = np.zeros((k + 1, k)); st = {}
t0 = time.time()
krylov.arnoldi_factorization(be, Q, H, 1, k, 0, stats=st)
print("%d Arnoldi steps of the linearised map about it: %.0f s per matvec (%d time steps), orthogonalisation %.2f s; Hessenberg diagonal %s" % (k, np.mean(st["matvec_s"]), h.nsteps, np.mean(st["orth_s"]), np.array2string(np.diag(H[:k, :k]), precision=5)))
sg = h.stats()
print("redone maps %d, capped solves %d" % (sg["retries"], sg["total_capped_solves"]))
h.close()
