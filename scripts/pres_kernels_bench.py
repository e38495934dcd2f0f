"""Back-to-back timings of the kernels of one pressure GMRES iteration and of the fixed part of a time step (config 2,
production settings): `nsk_bench_kernel` names coarse / schwarz / divgs / gmres_update / pres_chain / proj_apply / ...
NSK_LIB selects an experimental build for A/B runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, seed
from nekstab_amd.settings import production_context
G = os.path.join(ROOT, "tests", "golden")
case = mesh.load_case_npz(os.path.join(G, "cylinder_case.npz"), 8)
h = production_context(case)
q, f = h.alloc(2)
vx, vy = seed.add_noise(case)
h.upload(q, vx, vy, np.zeros(h.npres))
h.scal(q, 1.0 / h.norm(q))
for rep in range(3):
    t0 = time.time(); h.matvec(f, q, 0); n = h.norm(f); dt = time.time() - t0
    h.copy(q, f); h.scal(q, 1.0 / n)
st = h.stats()
print("matvec %.1f ms (%.1f Helmholtz + %.1f pressure iterations per step)" % (1e3 * dt, st["helm_iters"] / h.nsteps, st["pres_iters"] / h.nsteps))
names = sys.argv[1:] or ["helm", "coarse", "schwarz", "divgs", "gmres_update", "pres_chain", "rhs", "pres_rhs", "proj_apply", "pres_update", "vel_update_proj", "proj_update"]
for nm in names:
    r = h.bench_kernel(nm, 400)
    print("%-16s %7.2f us" % (nm, r["avg_us"]))
