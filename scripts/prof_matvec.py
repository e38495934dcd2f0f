"""Profiling driver: a few nsk_matvec calls on cfg 2 (cylinder, lx1=8) for rocprofv3."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nmat = int(sys.argv[2]) if len(sys.argv) > 2 else 3
th = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
tp = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
nproj = int(sys.argv[5]) if len(sys.argv) > 5 else 8
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=th, tol_pres=tp, tol_relative=1, nproj=nproj,
               schwarz_layers=2, max_helm_iter=100, max_pres_iter=48)
rng = np.random.default_rng(0)
u = rng.standard_normal((2,) + case.x.shape) * case.mask
vq, vf = h.alloc(2)
h.upload(vq, u[0], u[1], np.zeros(h.npres))
h.matvec(vf, vq, 0)
h.copy(vq, vf); h.scal(vq, 1.0 / h.norm(vq))
for k in range(nmat):
    t0 = time.time(); h.matvec(vf, vq, 0); dt = time.time() - t0
    print("matvec %d: %.4fs" % (k, dt), h.stats())
