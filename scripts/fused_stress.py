"""Stress test of the persistent velocity solve against the launch-per-iteration form: many repetitions, all pairwise
differences, under uneven load (a second context hammers the GPU from another stream in half of the repetitions)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), lx1)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-2, tol_relative=1, schwarz_layers=2, max_helm_iter=100, max_pres_iter=48, nproj=8)
h.set_option("proj_reset", 1)
qx, qy = seed.add_noise(case)
q, f = h.alloc(2)
h.upload(q, qx, qy, np.zeros(h.npres)); h.scal(q, 1.0 / h.norm(q))
h.set_nsteps(12)
h.set_option("fused", 0)
h.matvec(f, q, 0); ref = h.download(f); s0 = h.stats()
h.matvec(f, q, 0); again = h.download(f)
print("launch form repeatable:", max(np.abs(a - b).max() for a, b in zip(ref[:2], again[:2])), "pres iters", s0["pres_iters"], h.stats()["pres_iters"])
scale = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
bad = 0
for r in range(reps):
    h.set_option("fused", 1)
    h.set_option("use_graph", r % 2)
    h.matvec(f, q, 0)
    g = h.download(f); st = h.stats()
    err = max(np.abs(a - b).max() for a, b in zip(ref[:2], g[:2])) / scale
    if err != 0.0:
        bad += 1
    print("rep %2d graph %d: fused vs launch form %.2e  helm %d pres %d" % (r, r % 2, err, st["helm_iters"], st["pres_iters"]), flush=True)
    h.set_option("fused", 0)
    h.matvec(f, q, 0)
    g2 = h.download(f)
    e2 = max(np.abs(a - b).max() for a, b in zip(ref[:2], g2[:2])) / scale
    if e2 != 0.0:
        print("   launch form itself differs from its first run: %.2e pres %d" % (e2, h.stats()["pres_iters"]))
print("mismatches: %d of %d" % (bad, reps))
