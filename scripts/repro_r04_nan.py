#!/usr/bin/env python3
"""Reproduce the round-4 driver failure: NaN from nsk_orth on a vector made by nsk_matvec_batch after bench.py's diagnostics.

    python3 scripts/repro_r04_nan.py [budget|nobudget] [arn=N]

Sequence of BENCH_r04 (python3 bench.py --gpus 1 --steps 20 --warmup 5): 25 Arnoldi steps, continued to 128, bench_kernel of the
dominant kernel + the 11 step kernels (lane 0's live state), three extra contexts, then band Arnoldi b = 2, 3.  After every stage the
script checks that lane 0 still maps a unit vector to a finite one and reports which lane of matvec_batch goes non-finite."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from nekstab_amd import krylov, mesh, seed
from nekstab_amd.capi import NekStabHip, NskError
from nekstab_amd.settings import PRODUCTION, PRODUCTION_OPTIONS

budget = "nobudget" not in sys.argv
nocheck = "nocheck" in sys.argv
others = "others" in sys.argv
narn = 128
KERN = ["helm", "convect", "rhs", "pres_rhs", "proj_apply", "gmres_update", "schwarz", "divgs2", "pres_update", "vel_update_proj", "proj_update", "update_coarse3"]
for a in sys.argv[1:]:
    if a.startswith("arn="):
        narn = int(a[4:])
    if a.startswith("kern="):
        KERN = a[5:].split(",")
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=PRODUCTION["tol_helm"], tol_pres=PRODUCTION["tol_pres"], tol_relative=1,
               schwarz_layers=2, max_helm_iter=100, max_pres_iter=48, nproj=PRODUCTION["nproj"])
h.set_option("min_pres_iter", PRODUCTION_OPTIONS["min_pres_iter"])
qx, qy = seed.add_noise(case)
zp = np.zeros((case.nel, 6, 6))
Q = h.alloc(narn + 1)
h.upload(Q[0], qx, qy, zp)
h.scal(Q[0], 1.0 / h.norm(Q[0]))
H = np.zeros((narn + 1, narn))
krylov.arnoldi_factorization(h, Q, H, 1, narn, 0, stats={})
print("arnoldi done", narn, "stats", {k: h.stats()[k] for k in ("retries", "total_capped_solves")}, flush=True)


def check(tag, force=False):
    if nocheck and not force:
        return
    t = h.alloc(2)
    h.copy(t[0], Q[0])
    try:
        h.matvec(t[1], t[0], 0)
        print("  [%s] lane-0 map norm %.6e" % (tag, h.norm(t[1])), flush=True)
    except NskError as e:
        print("  [%s] lane-0 map FAILED: %s" % (tag, e), flush=True)
    h.free(t)


check("after arnoldi")
if budget:
    for kn in KERN:
        h.bench_kernel(kn, 200 if kn == "helm" else 100)
        check("after bench_kernel " + kn)
if others:
    for (th, tp, npj, opts) in ((1e-9, 3e-1, 8, {"min_pres_iter": 2, "pres_cap": 4}), (1e-11, 1e-1, 16, {"min_pres_iter": 2}), (PRODUCTION["tol_helm"], PRODUCTION["tol_pres"], 0, {"min_pres_iter": 2})):
        hc = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=th, tol_pres=tp, tol_relative=1, schwarz_layers=2, max_helm_iter=100, max_pres_iter=48, nproj=npj)
        for k, v in opts.items():
            hc.set_option(k, v)
        Qc = hc.alloc(30)
        hc.upload(Qc[0], qx, qy, zp)
        hc.scal(Qc[0], 1.0 / hc.norm(Qc[0]))
        krylov.arnoldi_factorization(hc, Qc, np.zeros((30, 29)), 1, 28, 0, stats={})
        hc.close()
    print("other contexts done", flush=True)
for bw in (2, 3):
    sd = h.alloc(bw)
    h.copy(sd[0], Q[0])
    for j in range(1, bw):
        h.upload(sd[j], qy * np.cos(0.2 * j * case.x), qx * np.cos(0.3 * j * case.y), zp)
    # by hand: the band loop with a finiteness check per lane
    Qb = h.alloc(48 + bw)
    for j in range(bw):
        h.copy(Qb[j], sd[j])
        h.orth(Qb[j], Qb[:j])
    i = 0
    bad = False
    while i < 48 and not bad:
        nb = min(bw, 48 - i)
        fs = [Qb[i + bw + j] for j in range(nb)]
        if nb > 1:
            h.matvec_batch(fs, [Qb[i + j] for j in range(nb)], 0)
        else:
            h.matvec(fs[0], Qb[i], 0)
        for j in range(nb):
            f = h.download(fs[j])
            fin = [bool(np.isfinite(np.asarray(c)).all()) for c in f]
            if not all(fin):
                print("  band b=%d column %d lane %d: non-finite fields %s" % (bw, i + j, j, fin), flush=True)
                bad = True
        if bad:
            break
        for j in range(nb):
            h.orth(fs[j], Qb[:i + bw + j])
        i += nb
    print("band b=%d: %s after %d columns; stats %s" % (bw, "NaN" if bad else "ok", i, {k: h.stats()[k] for k in ("retries", "total_capped_solves")}), flush=True)
    h.free(Qb); h.free(sd)
    check("after band %d" % bw, True)
