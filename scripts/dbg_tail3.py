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
    H = np.zeros((K + 1, K))
    krylov.arnoldi_factorization(h, Q, H, 1, K, 0, stats={})
    st = h.stats()
    h.close()
    return H, st


runs = {}
for name, tail, start, opts in (("base s1 a", 0, 1, ()), ("base s1 b", 0, 1, ()), ("tail2 s1 a", 2, 1, ()), ("tail2 s1 b", 2, 1, ()), ("base s0", 0, 0, ()), ("tail2 s0", 2, 0, ()),
                                ("tail2 s1 heads+30/+60", 1, 1, (("tail_off_h", 60), ("tail_off_p", 30)))):
    runs[name] = run(tail, start, opts=opts)
    print(name, "retries", runs[name][1]["retries"], "tail maps", runs[name][1]["tail_maps"], "budgets %.2f %.2f" % (runs[name][1]["step_budget_helm_mean"], runs[name][1]["step_budget_pres_mean"]), flush=True)
names = list(runs)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        print("%-24s vs %-24s max|dH| %.2e" % (names[i], names[j], np.abs(runs[names[i]][0] - runs[names[j]][0]).max()))
