"""Profile driver: a few steps of the z-extruded cylinder (hexahedral kernels), eager launches (NSK_USE_GRAPH=0)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 8
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
c3 = mesh3d.extrude_case(c2, nz, 0.5 * nz, periodic=True)
u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1) * c2.mask
h3 = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=2e-1, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=8)
if "PROJ_RESET" in os.environ: h3.set_option("proj_reset", int(os.environ["PROJ_RESET"]))
b0, b1 = h3.alloc(2)
rng = np.random.default_rng(0)
w = 1e-3 * rng.standard_normal(c3.x.shape) * c3.mask
h3.upload3(b0, mesh3d.extrude_field(u[0], nz), mesh3d.extrude_field(u[1], nz), w, np.zeros(h3.npres))
h3.set_nsteps(nst)
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
    t0 = time.time(); h3.matvec(b1, b0, 0); dt = time.time() - t0
    st = h3.stats()
    print("lx1 %d E %d pts/field %d: %.2f ms/step helm/step %.1f pres/step %.1f" % (lx1, c3.nel, h3.nvel, 1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst), flush=True)
    print(st, flush=True)
