import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_3d_gpu as T
c = T._case(6, True)
c.spng = 0.4 * np.clip(c.x - 1.4, 0.0, None) ** 2
h = T._hip(c, tol_pres=1e-8)
x, y, z = c.x, c.y, c.z
q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask, np.sin(x + y) * np.cos(2.0 * z) * c.mask]
p = np.zeros((c.nel, 4, 4, 4))
v0, v1 = h.alloc(2)
h.upload3(v0, *q, p); h.set_nsteps(5)
prev = None
for rep in range(1):
    for mode in (0, 1):
        try:
            h.matvec(v1, v0, mode)
        except Exception as e:
            print("ERR", e)
        out = h.download3(v1); st = h.stats()
        print(rep, mode, float(np.abs(out[0]).sum()), st["helm_iters"], st["pres_iters"], st["max_pres_iter"], st["unconverged"], st["budget_helm"], st["budget_pres"], st["retries"], flush=True)
    g = np.random.default_rng(4).standard_normal((c.nel, 4, 4, 4))
    xs, it = h.t_pres_solve(g); print("pres_solve iters", it, float(np.abs(xs).sum()))
