"""Config 4 at full size (backward-facing step extruded over 30 layers, E = 50 100, lx1 = 8, adjoint map) on one GPU:
time per step and a target for `rocprofv3 --kernel-trace --stats` (NSK_USE_GRAPH=0 for eager launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
G = os.path.join(ROOT, "tests", "golden")
c2 = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
c3 = mesh3d.extrude_case(c2, nz, 0.2 * nz, periodic=True)
t0 = time.time()
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=int(os.environ.get('MAXP', '48')), nproj=int(os.environ.get('NPROJ', '0')))
print("E %d set-up %.0f s nsteps %d" % (c3.nel, time.time() - t0, h.nsteps), flush=True)
tg = np.load(os.path.join(G, "backstep_tg.npz"))
u2 = mesh.interp_field_2d(tg["pRe_u"].astype(np.float64), 8) * c2.mask
w = 1e-2 * np.sin(2 * np.pi * c3.z / (0.2 * nz)) * c3.mask * np.abs(mesh3d.extrude_field(u2[0], nz))
q, f = h.alloc(2)
h.upload3(q, mesh3d.extrude_field(u2[0], nz), mesh3d.extrude_field(u2[1], nz), w, np.zeros(h.npres))
h.scal(q, 1.0 / h.norm(q))
h.set_nsteps(nst)
for rep in range(reps):
    t0 = time.time(); h.matvec(f, q, 1); h.norm(f); dt = time.time() - t0
    st = h.stats()
    print("%.1f ms per step (%.1f Helmholtz + %.1f pressure iterations per step)" % (1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst), flush=True)
