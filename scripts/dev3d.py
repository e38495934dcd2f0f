"""3-D development driver: iteration counts and parity of a few steps on the small test boxes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_3d_gpu as T
combos = [(6, False), (8, True)] if len(sys.argv) < 2 else [(int(sys.argv[1]), sys.argv[2] == "1")]
for lx1, outflow in combos:
    c = T._case(lx1, outflow)
    c.spng = 0.4 * np.clip(c.x - 1.4, 0.0, None) ** 2 if outflow else np.zeros_like(c.x)
    o = T._oracle(c)
    for tp in (1e-6,):
        h = T._hip(c, tol_pres=tp)
        x, y, z = c.x, c.y, c.z
        q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask, np.sin(x + y) * np.cos(2.0 * z) * c.mask]
        m = lx1 - 2
        p = np.zeros((c.nel, m, m, m))
        for mode in (0,):
            for nst in (1, 2, 5):
                ref = o.matvec((q[0], q[1], q[2], p), adjoint=bool(mode), nsteps=nst)
                v0, v1 = h.alloc(2)
                h.upload3(v0, *q, p); h.set_nsteps(nst)
                try:
                    h.matvec(v1, v0, mode)
                except Exception as e:
                    print("ERR", e)
                out = h.download3(v1); st = h.stats()
                sc = max(np.abs(ref[k]).max() for k in range(3))
                print(lx1, outflow, "tolp", tp, "mode", mode, "nst", nst, "err", [float(np.abs(out[k] - ref[k]).max() / sc) for k in range(3)],
                      "helm/step", st["helm_iters"] / nst, "pres/step", st["pres_iters"] / nst, "max", st["max_helm_iter"], st["max_pres_iter"], "unconv", st["unconverged"], flush=True)
        h.close()
