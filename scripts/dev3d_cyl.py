"""z-extruded cylinder: hexahedral path vs quadrilateral path for a z-invariant perturbation, and timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
c3 = mesh3d.extrude_case(c2, nz, 0.5 * nz, periodic=True)
u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1) * c2.mask
for tp, npj in ((1e-1, 8), (1e-4, 8), (1e-6, 0)):
    prod = dict(tol_helm=1e-11, tol_pres=tp, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=npj)
    h2 = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], **prod)
    t0 = time.time()
    h3 = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], **prod)
    t_init = time.time() - t0
    nst = 10
    a0, a1 = h2.alloc(2)
    h2.upload(a0, u[0], u[1], np.zeros(h2.npres)); h2.set_nsteps(nst); h2.matvec(a1, a0, 0)
    r2 = h2.download(a1)
    b0, b1 = h3.alloc(2)
    h3.upload3(b0, mesh3d.extrude_field(u[0], nz), mesh3d.extrude_field(u[1], nz), np.zeros(h3.nvel), np.zeros(h3.npres))
    h3.set_nsteps(nst)
    try:
        h3.matvec(b1, b0, 0)
    except Exception as e:
        print("ERR", e)
    t0 = time.time(); 
    try:
        h3.matvec(b1, b0, 0)
    except Exception as e:
        print("ERR", e)
    t3 = time.time() - t0
    r3 = h3.download3(b1)
    sc = np.abs(r2[0]).max()
    e = slice(0, c2.nel)
    st = h3.stats(); s2 = h2.stats()
    print("lx1", lx1, "nz", nz, "E3", c3.nel, "tolp", tp, "nproj", npj, "init3 %.1fs" % t_init, "err u %.2e v %.2e w %.2e" % (np.abs(r3[0][e, 1] - r2[0]).max() / sc, np.abs(r3[1][e, 2] - r2[1]).max() / sc, np.abs(r3[2]).max() / sc),
          "3D helm/step %.1f pres/step %.1f unconv %d  %.2f ms/step | 2D helm %.1f pres %.1f" % (st["helm_iters"] / nst, st["pres_iters"] / nst, st["unconverged"], 1e3 * t3 / nst, s2["helm_iters"] / nst, s2["pres_iters"] / nst), flush=True)
    h2.close(); h3.close()
