import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
qx, qy = seed.add_noise(case)
zp = np.zeros((case.nel, 6, 6))


def run(tail, start, K=6, opts=()):
    h = production_context(case)
    h.set_option("fuse2_start", start); h.set_option("tail", tail)
    for k, v in opts: h.set_option(k, v)
    Q = h.alloc(K + 1)
    h.upload(Q[0], qx, qy, zp)
    h.scal(Q[0], 1.0 / h.norm(Q[0]))
    recs = []
    H = np.zeros((K + 1, K))
    for m in range(1, K + 1):
        krylov.arnoldi_factorization(h, Q, H, m, m, 0, stats={})
        hh, pp = h.step_iters()
        recs.append((hh.copy(), pp.copy(), dict(h.stats())))
    h.close()
    return H, recs


for start in (1,):
    for name, opts in (("default", ()), ("step_budgets=0", (("step_budgets", 0),))):
        H0, r0 = run(0, start, opts=opts)
        H1, r1 = run(2, start, opts=opts)
        print("start", start, name, "H equal", np.array_equal(H0, H1))
        for m, (a, b) in enumerate(zip(r0, r1)):
            dh = np.nonzero(a[0] != b[0])[0]; dp = np.nonzero(a[1] != b[1])[0]
            print("  map %d: retries %d / %d, tail maps %d, first differing step: helm %s pres %s" % (m + 1, a[2]["retries"], b[2]["retries"], b[2]["tail_maps"], dh[:1], dp[:1]),
                  "" if len(dp) == 0 else "pres counts around it: %s vs %s" % (a[1][max(0, dp[0] - 2):dp[0] + 3], b[1][max(0, dp[0] - 2):dp[0] + 3]))
