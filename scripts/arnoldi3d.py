"""Arnoldi on the z-extruded cylinder (hexahedral path, 3-D noise seed): the spectrum of the spanwise-periodic problem
contains the 2-D one, and at Re = 50 the leading pair is the 2-D Hopf pair of Spectre_Hd.dat (0.7387113 +- 0.6972442i)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, mesh3d, seed
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kdim = int(sys.argv[3]) if len(sys.argv) > 3 else 80
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
c3 = mesh3d.extrude_case(c2, nz, 1.0 * nz, periodic=True)
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=2e-1, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=48)
if "PROJ_RESET" in os.environ: h.set_option("proj_reset", int(os.environ["PROJ_RESET"]))
qx, qy = seed.add_noise(c2)
rng = np.random.default_rng(1)
zz = c3.z
q3 = [mesh3d.extrude_field(qx, nz) * (1 + 0.3 * np.sin(2 * np.pi * zz / nz)), mesh3d.extrude_field(qy, nz) * (1 + 0.3 * np.cos(2 * np.pi * zz / nz)),
      0.3 * mesh3d.extrude_field(qx, nz) * np.sin(2 * np.pi * zz / nz)]
v0 = h.alloc(1)[0]
h.upload3(v0, q3[0] * c3.mask, q3[1] * c3.mask, q3[2] * c3.mask, np.zeros(h.npres))
t0 = time.time()
def log(m, H, dt):
    if m % 10 == 0:
        vals, vecs = krylov.eig_sorted(H[:m, :m]); r = np.abs(H[m, m - 1] * vecs[m - 1, :])
        print("k=%3d  %.2fs/iter  leading %.7f %+.7fi  residual %.2e" % (m, dt, vals[0].real, vals[0].imag, r[0]), flush=True)
res = krylov.krylov_schur(h, v0, kdim, mode=0, schur_tgt=0, log=log)
print("E=%d lx1=%d: %d matvecs in %.1fs; leading Ritz values:" % (c3.nel, lx1, res.matvecs, time.time() - t0))
for i in range(6):
    print("  %.7f %+.7fi  |mu| %.6f  residual %.2e" % (res.vals[i].real, res.vals[i].imag, abs(res.vals[i]), res.residual[i]))
