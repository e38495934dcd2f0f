import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from nekstab_amd.capi import NekStabHip
from nekstab_amd.sharded import ShardGroup
from tests.test_sharded_r3_gpu import _cavity, _rel
c2, p2 = _cavity()
for tp, mp in ((1e-4, 48), (1e-7, 192)):
    h = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], tol_helm=1e-12, tol_pres=tp, tol_relative=1, nproj=0, max_helm_iter=150, max_pres_iter=mp)
    bump = 1e-2 * np.sin(np.pi * c2.x) * np.sin(np.pi * c2.y / 1.2) * c2.mask
    q = (c2.ub[0] + bump, c2.ub[1] - 0.5 * bump, p2)
    w = np.ones_like(c2.x)
    for nr in (1, 3):
        g = ShardGroup(h, c2, nr)
        vq, vf = h.alloc(2); sq, sf = g.alloc(2)
        h.upload(vq, *q); g.upload(sq, *q)
        for ns in (1, 2, 12):
            h.set_nsteps(ns); g.set_nsteps(ns)
            for mode in ("lin", "nl"):
                if mode == "lin":
                    h.matvec(vf, vq, 0); s1 = h.stats(); g.matvec(sf, sq, 0)
                else:
                    h.nonlinear_map(vf, vq); s1 = h.stats(); g.nonlinear_map(sf, sq)
                s2 = g.stats()
                print(tp, "ranks", nr, "steps", ns, mode, "rel", _rel(w, g.download(sf), h.download(vf)), "its", s1["helm_iters"], s1["pres_iters"], s1["max_pres_iter"], "|", s2["helm_iters"], s2["pres_iters"], s2["max_pres_iter"], flush=True)
        g.close()
    h.close()
