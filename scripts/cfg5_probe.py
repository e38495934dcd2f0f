"""Config 5 at full size (46 x 46 x 47 hexahedra, lx1 = 10): which inner solve needs how many iterations on the stretched box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NSK_DEBUG"] = "1"
import numpy as np
from nekstab_amd import mesh3d
from nekstab_amd.capi import NekStabHip, NskError
n = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else [46, 46, 47]
beta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0          # 1: Chebyshev spacing (cav.box), 0: uniform
emul = float(sys.argv[5]) if len(sys.argv) > 5 else 0.01
stretch = lambda xi: (1.0 - beta) * xi + beta * 0.5 * (1.0 - np.cos(np.pi * xi))
c = mesh3d.box_case_3d(n[0], n[1], n[2], 10, lengths=(1.0, 1.0, 1.0), re=1000.0, endtime=0.02, stretch=stretch)
sx, sy, sz = np.sin(np.pi * c.x), np.sin(np.pi * c.y), np.sin(np.pi * c.z)
c.ub[0] = sx ** 2 * np.sin(2 * np.pi * c.y) * sz ** 2 * c.mask
c.ub[1] = -np.sin(2 * np.pi * c.x) * sy ** 2 * sz ** 2 * c.mask
del sx, sy, sz
t0 = time.time()
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-9, tol_pres=1e-2, tol_relative=1, max_helm_iter=400, max_pres_iter=192)
print("set-up %.0f s, nsteps %d dt %.3e" % (time.time() - t0, h.nsteps, h.dt), flush=True)
h.set_option("use_graph", 0)
h.set_option("early_pres_mul", emul)
ed = np.diff(stretch(np.linspace(0, 1, n[0] + 1)))
print("beta %.2f early_pres_mul %g: cell size ratio max/min %.1f" % (beta, emul, ed.max() / ed.min()), flush=True)
q, f = h.alloc(2)
rng = np.random.default_rng(2)
w = 1e-2 * rng.standard_normal(c.x.shape) * c.mask
h.upload3(q, c.ub[0] + w, c.ub[1] - w, w, np.zeros(h.npres))
h.set_nsteps(3)
for mode in ("lin",):
    try:
        t0 = time.time()
        (h.matvec(f, q, 0) if mode == "lin" else h.nonlinear_map(f, q))
        print(mode, "ok %.1f s" % (time.time() - t0), h.stats(), flush=True)
    except NskError as e:
        print(mode, "FAILED", e, h.stats(), flush=True)
