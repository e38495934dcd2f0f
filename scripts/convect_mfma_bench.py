"""Hexahedral convection kernel, thread-per-node (LDS FMA loops) against the matrix-core form (nsk3_mfma.hpp):
HIP-event time per launch on the z-extruded cylinder (E = 3992, lx1 = 8: 2.04 M points per field), and the time per step
of a short map with either kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
c3 = mesh3d.extrude_case(c2, nz, 0.5 * nz, periodic=True)
u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), 8) * c2.mask
h3 = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=2e-1, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=8)
b0, b1 = h3.alloc(2)
rng = np.random.default_rng(0)
w = 1e-3 * rng.standard_normal(c3.x.shape) * c3.mask
h3.upload3(b0, mesh3d.extrude_field(u[0], nz), mesh3d.extrude_field(u[1], nz), w, np.zeros(h3.npres))
h3.set_nsteps(12)
for mf in (1, 0, 1, 0):
    h3.set_option("mfma_convect", mf)
    h3.matvec(b1, b0, 0)
    t0 = time.time(); h3.matvec(b1, b0, 0); h3.norm(b1); dt = time.time() - t0
    print("mfma_convect %d: %.3f ms per step" % (mf, 1e3 * dt / 12), h3.stats()["helm_iters"], flush=True)
a = h3.bench_kernel("convect", 50); b = h3.bench_kernel("convect_mfma", 50)
P = h3.nvel
print("E %d points/field %d: k_convect<8> %.1f us, k_convect_mfma8 %.1f us per launch (%.2fx); algorithmic 176 B/pt => %.2f / %.2f TB/s"
      % (c3.nel, P, a["avg_us"], b["avg_us"], a["avg_us"] / b["avg_us"], 176.0 * P / a["avg_us"] / 1e6, 176.0 * P / b["avg_us"] / 1e6))
