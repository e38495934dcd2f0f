"""Config 3 (cylinder, 2 x 2 refined mesh, E = 7984, lx1 = 12) on one GPU: time per step in the body of a map, and a target for
`rocprofv3 --kernel-trace --stats -- python3 scripts/prof_cfg3.py` (NSK_USE_GRAPH=0 for eager launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, seed
from nekstab_amd.settings import production_context
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 60
case = mesh.refine_case_2x2(mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 12))
h = production_context(case)
qx, qy = seed.add_noise(case)
q, f = h.alloc(2)
h.upload(q, qx, qy, np.zeros(h.npres)); h.scal(q, 1.0 / h.norm(q))
h.set_nsteps(nst)
for rep in range(3):
    t0 = time.time(); h.matvec(f, q, 0); h.norm(f); dt = time.time() - t0
    st = h.stats()
    print("E %d lx1 12: %.2f ms per step over %d steps (%.1f Helmholtz + %.1f pressure iterations per step)" % (case.nel, 1e3 * dt / nst, nst, st["helm_iters"] / nst, st["pres_iters"] / nst), flush=True)
    h.copy(q, f); h.scal(q, 1.0 / h.norm(q))
