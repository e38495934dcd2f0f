"""Config 5 (stretched box of hexahedra, lx1 = 10): per-kernel profiling target.  Default 24^3 elements (13 824 elements,
13.8 M points per field) so that a kernel trace stays small; `46 46 47` is the full size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh3d
from nekstab_amd.capi import NekStabHip
n = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else [24, 24, 24]
nst = int(sys.argv[4]) if len(sys.argv) > 4 else 3
stretch = lambda xi: 0.5 * (1.0 - np.cos(np.pi * xi))
c = mesh3d.box_case_3d(n[0], n[1], n[2], 10, lengths=(1.0, 1.0, 1.0), re=1000.0, endtime=0.02, stretch=stretch)
sx, sy, sz = np.sin(np.pi * c.x), np.sin(np.pi * c.y), np.sin(np.pi * c.z)
c.ub[0] = sx ** 2 * np.sin(2 * np.pi * c.y) * sz ** 2 * c.mask
c.ub[1] = -np.sin(2 * np.pi * c.x) * sy ** 2 * sz ** 2 * c.mask
del sx, sy, sz
t0 = time.time()
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-9, tol_pres=1e-2, tol_relative=1, max_helm_iter=400, max_pres_iter=192, nproj=int(os.environ.get('NPROJ', '0')))
print("E %d set-up %.0f s, nsteps %d dt %.3e" % (c.nel, time.time() - t0, h.nsteps, h.dt), flush=True)
if os.environ.get("MFMA_CONVECT") is not None:           # 0: the thread-per-node convection kernels (A/B of the matrix-core ones)
    h.set_option("mfma_convect", int(os.environ["MFMA_CONVECT"]))
q, f = h.alloc(2)
rng = np.random.default_rng(2)
if os.environ.get("SMOOTH", "0") == "1":      # a smooth, continuous, three-dimensional perturbation (what a Krylov vector of a physical run looks like after a few maps)
    w = 1e-2 * np.sin(2 * np.pi * c.x) * np.sin(3 * np.pi * c.y) * np.sin(2 * np.pi * c.z) * c.mask
else:                                          # node-wise noise: discontinuous across elements, the hardest input there is
    w = 1e-2 * rng.standard_normal(c.x.shape) * c.mask
h.upload3(q, c.ub[0] + w, c.ub[1] - w, w, np.zeros(h.npres))
h.set_nsteps(nst)
for rep in range(int(os.environ.get('REPS', '2'))):
    t0 = time.time()
    try:
        if os.environ.get("MODE") == "nl":                 # the FULL equations' map (Newton-Krylov's nonlinear step) instead of the linearised one
            h.nonlinear_map(f, q); h.norm(f)
        else:
            h.matvec(f, q, 0); h.norm(f)
    except Exception as exc:                               # noqa: BLE001  (timing experiments with wrong values)
        print("note:", exc)
    dt = time.time() - t0
    st = h.stats()
    print("%.1f ms per step (%.1f Helmholtz + %.1f pressure iterations per step)" % (1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst), flush=True)
for kn in os.environ.get("KERNELS", "").split():          # HIP-event timings of single kernels on the state the last map left
    try:
        print("%-16s %9.1f us" % (kn, h.bench_kernel(kn, 10)["avg_us"]), flush=True)
    except Exception as exc:                               # noqa: BLE001
        print("%-16s %s" % (kn, exc))
print("zero_arrays = 0x%x" % h.stats().get("zero_arrays", 0))
