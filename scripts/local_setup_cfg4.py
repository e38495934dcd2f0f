#!/usr/bin/env python3
"""Rank-local set-up at BASELINE configs[3]'s size (backward-facing step over 30 spanwise layers, E = 50 100, lx1 = 8): R virtual
ranks on one GPU, each set up from ITS sub-mesh (own elements + two rings: sharded.LocalParent), coarse rows gathered, the
block-circulant coarse solve detected on the gathered rows; a short adjoint map on the R shards against the same map on the
whole-mesh context.  Prints the set-up time and the elements per rank next to the whole-mesh set-up.

    python scripts/local_setup_cfg4.py [R=4] [steps=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
from nekstab_amd.sharded import LocalParent, ShardGroup, partition_rcb
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nz = 30
G = os.path.join(ROOT, "tests", "golden")
c2 = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
c3 = mesh3d.extrude_case(c2, nz, 0.2 * nz, periodic=True)
KW = dict(tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=96, nproj=0)
tg = np.load(os.path.join(G, "backstep_tg.npz"))
u2 = mesh.interp_field_2d(tg["pRe_u"].astype(np.float64), 8) * c2.mask
w = 1e-1 * np.sin(2 * np.pi * c3.z / (0.2 * nz)) * c3.mask * np.abs(mesh3d.extrude_field(u2[0], nz))
q = (mesh3d.extrude_field(u2[0], nz), mesh3d.extrude_field(u2[1], nz), w, np.zeros((c3.nel, 6, 6, 6)))
part = partition_rcb(c3, R)
P, tl = [], []
for r in range(R):
    t0 = time.time()
    P.append(LocalParent(c3, part, r, **KW))
    tl.append(time.time() - t0)
    print("rank %d: %d own + %d ring elements of %d, local set-up %.1f s, %d coarse-row entries" % (r, int((part == r).sum()), P[-1].nel - int((part == r).sum()), c3.nel, tl[-1], len(P[-1].rows_a)), flush=True)
t0 = time.time()
vol, npr = sum(p.vol_own for p in P), sum(p.npr_own for p in P)
ct, lm = max(p.ctarg for p in P), max(p.fd_lmax for p in P)
u, v, a = (np.concatenate([getattr(p, k) for p in P]) for k in ("rows_u", "rows_v", "rows_a"))
tf = []
for p in P:
    t1 = time.time(); p.finish(vol, ct, lm, npr, u, v, a); tf.append(time.time() - t1)
print("coarse rows of all ranks: %d entries (%.0f MB); finish (dt, Jacobi diagonals, replicated coarse solve) %.1f s per rank" % (len(a), 16e-6 * len(a), np.mean(tf)), flush=True)
t0 = time.time()
g = ShardGroup(P, c3, R, part)
print("cutting the %d shards (halo tables, slices of the parents' arrays, work arrays): %.1f s" % (R, time.time() - t0), flush=True)
g.release_parent()
for k, v in (("shard_hostcheck", os.environ.get("HOSTCHECK")), ("halo_overlap", os.environ.get("HALO_OVERLAP"))):
    if v is not None:
        g.set_option(k, int(v))
g.set_nsteps(nst)
sq, sf = g.alloc(2)
g.upload3(sq, *q)
g.matvec(sf, sq, 1)                                   # (first map: captures / settles)
t0 = time.time(); g.matvec(sf, sq, 1); g.norm(sf); tm = time.time() - t0
got = g.download3(sf)
sl = g.stats()
print("R = %d shards (virtual ranks, one GPU): %d adjoint steps in %.2f s, %.1f Helmholtz + %.1f pressure iterations per step" % (R, nst, tm, sl["helm_iters"] / nst, sl["pres_iters"] / nst), flush=True)
g.close()
for p in P:
    p.close()
t0 = time.time()
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], **KW)
tw = time.time() - t0
ns_whole = h.nsteps
h.set_nsteps(nst)
vq, vf = h.alloc(2)
h.upload3(vq, *q)
t0 = time.time(); h.matvec(vf, vq, 1); h.norm(vf); tm1 = time.time() - t0
ref = h.download3(vf)
sw = h.stats()
sc = max(np.abs(ref[k]).max() for k in range(3))
err = max(np.abs(got[k] - ref[k]).max() for k in range(3)) / sc
print("whole-mesh context: set-up %.1f s (rank-local: %.1f s per rank + %.1f s finish), %d steps in %.2f s, %.1f + %.1f iterations per step" % (tw, np.mean(tl), np.mean(tf), nst, tm1, sw["helm_iters"] / nst, sw["pres_iters"] / nst))
print("velocity difference local shards vs whole-mesh single rank: %.2e (nsteps per map: local %d, whole mesh %d; dt equal: %s)" % (err, P[0].nsteps, ns_whole, abs(P[0].dt - h.dt) < 1e-15))
h.close()
