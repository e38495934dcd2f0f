"""k_dim=200 Arnoldi on the GPU (lx1=6 cylinder) vs the reference's Spectre_Hd.dat, row by row."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.capi import NekStabHip
th = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-11
tp = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 6)
sp = np.load(os.path.join(ROOT, "tests/golden/cylinder_spectre.npz"))["Hd"]
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=th, tol_pres=tp, tol_relative=1, nproj=8, max_helm_iter=120, max_pres_iter=48)
qx, qy = seed.add_noise(case)
v0, v1 = h.alloc(2)
h.upload(v0, qx, qy, np.zeros(h.npres)); h.scal(v0, 1.0 / h.norm(v0)); h.matvec(v1, v0, 0)
res = krylov.krylov_schur(h, v1, 200, schur_tgt=0)
print("wall", res.wall)
for r in sp[sp[:, 2] < 1e-6]:
    z = complex(r[0], r[1]); j = np.argmin(np.abs(res.vals - z))
    print("ref %.7f%+.7fi res %.1e | ours %.9f%+.9fi res %.1e | diff %.2e" % (z.real, z.imag, r[2], res.vals[j].real, res.vals[j].imag, res.residual[j], abs(res.vals[j] - z)))
