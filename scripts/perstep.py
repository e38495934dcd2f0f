"""cfg 2: per-step iteration counts of one map (NSK_DEBUG=1 NSK_USE_GRAPH=0 prints them on stderr)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.capi import NekStabHip
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-10, tol_pres=2e-1, tol_relative=1, nproj=8, schwarz_layers=2, max_helm_iter=100, max_pres_iter=48)
if os.environ.get("PRES_FLOOR"): h.set_option("pres_floor", float(os.environ["PRES_FLOOR"]))
qx, qy = seed.add_noise(c)
Q = h.alloc(12)
h.upload(Q[0], qx, qy, np.zeros(h.npres)); h.scal(Q[0], 1.0 / h.norm(Q[0]))
H = np.zeros((12, 11))
krylov.arnoldi_factorization(h, Q, H, 1, 10, 0)
