#!/usr/bin/env python3
"""BASELINE configs[2] as an EIGENPROBLEM at full size on one GPU (until round 3 the full-size runs were
property checks over a few time steps): cylinder Re = 50 on the 2 x 2 refined mesh, E = 7984, lx1 = 12, 861 time steps per
matvec, direct Krylov-Schur with k_dim and schur_tgt from the command line, production settings.  Prints the restart log, the
converged eigenvalues next to the lx1 = 8 values (same operator, coarser discretisation) and the per-matvec cost.

    python scripts/run_cfg3_eigen.py [k_dim=64] [schur_tgt=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
k_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tgt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
case = mesh.refine_case_2x2(mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 12))
h = production_context(case, max_helm_iter=250, max_pres_iter=144)      # (lx1 = 12: the first time steps of a map can need more than the 100 iterations config 2 is capped at;
                                                        #  and the tightened pressure solves of steps 1-3 more than one 48-vector GMRES cycle; with host-read convergence flags a
                                                        #  high cap costs nothing, a cap that is hit costs a redone map: the '5 redone maps' of round 3 were cap hits)
print("E %d lx1 %d: %d points per field, state %d, nsteps %d, set-up %.0f s" % (case.nel, case.lx1, h.nvel, h.nstate, h.nsteps, time.time() - t0), flush=True)
qx, qy = seed.add_noise(case)
v0, v1 = h.alloc(2)
h.upload(v0, qx, qy, np.zeros(h.npres))
h.scal(v0, 1.0 / h.norm(v0))
h.matvec(v1, v0, 0)
tl = [time.time()]

def log(m, H, dt):
    if m % 8 == 0:
        vals, vecs = krylov.eig_sorted(H[:m, :m])
        res = abs(H[m, m - 1] * vecs[m - 1, 0])
        print("  step %3d  %.2f s per Arnoldi step  leading %.8f%+.8fi  residual %.1e" % (m, dt, vals[0].real, abs(vals[0].imag), res), flush=True)

t0 = time.time()
res = krylov.krylov_schur(h, v1, k_dim, schur_tgt=tgt, log=log)
wall = time.time() - t0
st = h.stats()
print("Krylov-Schur: k_dim %d, schur_tgt %d: %d matvecs, %d restarts, %.0f s (%.2f s per matvec; %.2f Helmholtz + %.2f pressure iterations per time step, %d redone maps)" % (
    k_dim, tgt, res.matvecs, res.schur_cnt, wall, wall / res.matvecs, st["total_helm_iters"] / st["total_steps"], st["total_pres_iters"] / st["total_steps"], st["retries"]))
lam = krylov.log_transform(res.vals, case.endtime)
for i in range(len(res.vals)):
    if res.residual[i] < 1e-6:
        print("  converged: mu = %.8f%+.8fi  lambda = %.7f%+.7fi  residual %.1e" % (res.vals[i].real, res.vals[i].imag, lam[i].real, lam[i].imag, res.residual[i]))
print("  lx1 = 8 (config 2), converged solves: mu = 0.73868738+0.69723066i; reference tables: direct lx1 = 6 0.7387113+0.6972442i, adjoint lx1 = 8 0.7386891-0.6972319i")
