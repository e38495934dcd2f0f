import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from nekstab_amd import mesh, seed
from nekstab_amd.capi import NekStabHip
from nekstab_amd.sharded import ShardGroup
adj = int(sys.argv[1]) if len(sys.argv) > 1 else 1
case = mesh.load_case_npz("tests/golden/cylinder_case.npz", 6, adjoint=bool(adj))
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=0, max_helm_iter=150, max_pres_iter=48)
h.set_option("use_graph", 0)
qx, qy = seed.add_noise(case)
for ns in (2,):
    h.set_nsteps(ns)
    vq, vf = h.alloc(2)
    h.upload(vq, qx, qy, np.zeros(h.npres))
    h.matvec(vf, vq, adj)
    print("single", ns, {k: v for k, v in h.stats().items() if k in ("helm_iters", "pres_iters", "max_helm_iter", "max_pres_iter", "unconverged")}, flush=True)
    ref = h.download(vf)
    for nr in (1, 2):
        g = ShardGroup(h, case, nr)
        g.set_option("shard_graph", 0)
        g.set_nsteps(ns)
        sq, sf = g.alloc(2)
        g.upload(sq, qx, qy, np.zeros(h.npres))
        try:
            g.matvec(sf, sq, adj)
            got = g.download(sf)
            print(" shards", nr, {k: v for k, v in g.stats().items() if k in ("helm_iters", "pres_iters", "max_helm_iter", "max_pres_iter", "unconverged")}, "diff", max(np.abs(a - b).max() for a, b in zip(got, ref)), flush=True)
        except Exception as e:
            print(" shards", nr, "FAILED", str(e)[:400], flush=True)
        g.close()
