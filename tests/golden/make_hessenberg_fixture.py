#!/usr/bin/env python3
"""Generates tests/golden/cfg2_hessenberg.npz (run on a GPU box; writes gpurun_out/cfg2_hessenberg.npz, copy it here):
the Hessenberg matrices of a Krylov-Schur run of BASELINE configs[1] (cylinder Re = 50, lx1 = 8, E = 1996, production
inner-solver settings) on the HIP path with k_dim = 40, schur_tgt = 2 -- H(k+1, k) as it stands when the first and the second
restart are taken (input of schur_condensation, core/eigensolvers.f:395-499) and at the end.  Inputs of
tests/test_dense_ref.py, which runs the reference's own eig / schur / ordschur (oracle/_ref) on them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context

case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
h = production_context(case)
qx, qy = seed.add_noise(case)
v0 = h.alloc(1)[0]
h.upload(v0, qx, qy, np.zeros(h.npres))
K = 40
snaps = []
orig = krylov.schur_condensation


def tap(be, Q, H, k, mstart, schur_del, schur_tgt):
    snaps.append(H.copy())
    return orig(be, Q, H, k, mstart, schur_del, schur_tgt)


krylov.schur_condensation = tap
res = krylov.krylov_schur(h, v0, K, mode=0, schur_tgt=2, eigen_tol=1e-6, max_restarts=6)
out = os.path.join(ROOT, "gpurun_out", "cfg2_hessenberg.npz")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez(out, k_dim=K, schur_tgt=2, schur_del=0.10, H_restart=np.array(snaps), H_final=res.H, vals_final=res.vals, residual_final=res.residual,
         provenance="HIP path, cylinder Re=50 lx1=8 E=1996, production settings, krylov_schur(k_dim=40, schur_tgt=2, eigen_tol=1e-6); tests/golden/make_hessenberg_fixture.py")
print("restarts", res.schur_cnt, "matvecs", res.matvecs, "leading", res.vals[:2], res.residual[:2])
