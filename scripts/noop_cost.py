#!/usr/bin/env python3
"""What does a launch that finds its solve converged cost in the un-profiled, graph-replayed step?
Config 2 at the production settings: the launch budgets are settled on a late Krylov vector, frozen, and the same map is
timed with the budgets as they are, with 8 and 16 more velocity-solve launches per step and with 4 and 8 more pressure
iterations (3 kernels each) per step -- every extra launch is a no-op by construction.  Graph replay and eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
for graph in (1, 0):
    h = production_context(case)
    h.set_option("use_graph", graph)
    qx, qy = seed.add_noise(case)
    Q = h.alloc(14)
    h.upload(Q[0], qx, qy, np.zeros(h.npres))
    h.scal(Q[0], 1.0 / h.norm(Q[0]))
    H = np.zeros((14, 13))
    krylov.arnoldi_factorization(h, Q, H, 1, 12, 0)            # settles the budgets (window of 8 maps)
    st = h.stats()
    f = h.alloc(1)[0]
    h.set_option("budget_freeze", 1)

    def timed(nrep=6):
        h.matvec(f, Q[12], 0); h.norm(f)
        t0 = time.perf_counter()
        for _ in range(nrep):
            h.matvec(f, Q[12], 0)
        h.norm(f)
        return (time.perf_counter() - t0) / nrep / h.nsteps * 1e6

    base = timed()
    s0 = h.stats()
    print("%s: budgets helm %d pres %d (tail class), %.2f + %.2f iterations per step in this map, %.1f us per time step" % (
        "graph replay" if graph else "eager launches", st["budget_helm"], st["budget_pres"], s0["helm_iters"] / h.nsteps, s0["pres_iters"] / h.nsteps, base), flush=True)
    prev = base
    for add in (8, 8):
        h.set_option("budget_add_helm", add)
        t = timed()
        print("   + %d k_helm launches per step: %.1f us per step -> %.2f us per no-op launch" % (add, t, (t - prev) / add), flush=True)
        prev = t
    h.set_option("budget_add_helm", -16)
    prev = timed()
    for add in (4, 4):
        h.set_option("budget_add_pres", add)
        t = timed()
        print("   + %d pressure iterations (3 launches each) per step: %.1f us per step -> %.2f us per no-op launch" % (add, t, (t - prev) / add / 3), flush=True)
        prev = t
    h.close()
