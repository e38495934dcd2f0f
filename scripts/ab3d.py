"""A/B timing of hexahedral builds (NSK_LIB=<lib>): full-work k_helm launch and ms per time step
on the z-extruded cylinder (E = 3992, lx1 = 8)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
c3 = mesh3d.extrude_case(c2, nz, 0.5 * nz, periodic=True)
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=2e-1, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=8)
rng = np.random.default_rng(0)
v0, v1 = h.alloc(2)
h.upload3(v0, *(rng.standard_normal(c3.x.shape) * c3.mask for _ in range(3)), np.zeros(h.npres))
ns = 24
h.set_nsteps(ns)
h.matvec(v1, v0, 0); h.matvec(v1, v0, 0)
t0 = time.perf_counter(); h.matvec(v1, v0, 0); t1 = time.perf_counter()
f = h.download3(v1)
st = h.stats()
print("lib", os.environ.get("NSK_LIB", "default"), "helm_us", round(h.bench_kernel("helm", 100)["avg_us"], 1),
      "ms/step", round(1e3 * (t1 - t0) / ns, 3), "helm_it", st["helm_iters"], "pres_it", st["pres_iters"],
      "chk", repr(float(np.sum(f[0] ** 2) + np.sum(f[1] ** 2))), flush=True)
