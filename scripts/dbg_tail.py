"""Which tail breaks bit-identity?  Launch budgets against tails with the heads of ONE solve pushed below / far above the counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
qx, qy = seed.add_noise(case)
zp = np.zeros((case.nel, 6, 6))
K = 6


def run(fuse2, tail, off_h=0, off_p=0, start=1):
    h = production_context(case)
    h.set_option("fuse2", fuse2); h.set_option("fuse2_start", start)
    h.set_option("tail", tail); h.set_option("tail_off_h", off_h); h.set_option("tail_off_p", off_p)
    Q = h.alloc(K + 1)
    h.upload(Q[0], qx, qy, zp)
    h.scal(Q[0], 1.0 / h.norm(Q[0]))
    H = np.zeros((K + 1, K))
    krylov.arnoldi_factorization(h, Q, H, 1, K, 0, stats={})
    st = h.stats(); hh, pp = h.step_iters()
    h.close()
    return H, hh.copy(), pp.copy(), st


for fuse2 in (1, 0):
    for start in ((1, 0) if fuse2 else (0,)):
        H0, h0, p0, s0 = run(fuse2, 0, start=start)
        for name, tail, oh, op in (("both tails, median heads", 1, 0, 0), ("velocity tail works, pressure heads +30", 1, -6, 30), ("pressure tail works, velocity heads +60", 1, 60, -3), ("safety net", 2, 0, 0)):
            H1, h1, p1, s1 = run(fuse2, tail, oh, op, start=start)
            print("fuse2 %d start %d %-44s H equal %s  helm counts equal %s  pres counts equal %s  max|dH| %.2e  tail maps %d" % (fuse2, start, name, np.array_equal(H0, H1), np.array_equal(h0, h1), np.array_equal(p0, p1), np.abs(H0 - H1).max(), s1["tail_maps"]), flush=True)
