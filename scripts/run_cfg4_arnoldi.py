#!/usr/bin/env python3
"""BASELINE configs[3] as an Arnoldi factorisation at full size on one GPU: backward-facing step extruded over 30 spanwise layers
(E = 50 100 hexahedra, lx1 = 8, 25.65 M points per field, state vector 702 MB), ADJOINT map (315 time steps), k Arnoldi steps from
a three-dimensional seed; prints the cost per matvec, the iteration counts and the Ritz values.

    python scripts/run_cfg4_arnoldi.py [k=12] [nproj=32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh, mesh3d
from nekstab_amd.capi import NekStabHip
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nproj = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nz = 30
G = os.path.join(ROOT, "tests", "golden")
c2 = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
c3 = mesh3d.extrude_case(c2, nz, 0.2 * nz, periodic=True)
t0 = time.time()
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=96, nproj=nproj)
print("E %d lx1 8: %d points per field, state %.0f MB, nsteps %d, set-up %.0f s" % (c3.nel, h.nvel, 8e-6 * h.nstate, h.nsteps, time.time() - t0), flush=True)
tg = np.load(os.path.join(G, "backstep_tg.npz"))
u2 = mesh.interp_field_2d(tg["pRe_u"].astype(np.float64), 8) * c2.mask
w = 1e-1 * np.sin(2 * np.pi * c3.z / (0.2 * nz)) * c3.mask * np.abs(mesh3d.extrude_field(u2[0], nz))
Q = h.alloc(k + 1)
h.upload3(Q[0], mesh3d.extrude_field(u2[0], nz), mesh3d.extrude_field(u2[1], nz), w, np.zeros(h.npres))
h.scal(Q[0], 1.0 / h.norm(Q[0]))
H = np.zeros((k + 1, k))
st = {}
def log(m, Hm, dt):
    s = h.stats()
    print("  Arnoldi step %2d: %.1f s (%.1f ms per time step, %.1f Helmholtz + %.1f pressure iterations per step)" % (m, dt, 1e3 * dt / h.nsteps, s["helm_iters"] / h.nsteps, s["pres_iters"] / h.nsteps), flush=True)
t0 = time.time()
krylov.arnoldi_factorization(h, Q, H, 1, k, 1, log=log, stats=st)
wall = time.time() - t0
vals, vecs = krylov.eig_sorted(H[:k, :k])
res = np.abs(H[k, k - 1] * vecs[k - 1, :])
print("%d adjoint matvecs in %.0f s = %.1f s per matvec (%.1f ms per time step); orthogonalisation %.2f s per step" % (k, wall, wall / k, 1e3 * np.mean(st["matvec_s"]) / h.nsteps, np.mean(st["orth_s"])))
print("Ritz values:", " ".join("%.4f%+.4fi(%.0e)" % (v.real, v.imag, r) for v, r in zip(vals[:6], res[:6])))
sg = h.stats()
print("redone maps %d, capped solves %d" % (sg["retries"], sg["total_capped_solves"]))
