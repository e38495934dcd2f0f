"""Converged direct spectra of the cylinder case FROM THE CPU ORACLE (not from the HIP path): the data
``tests/test_spectrum_pin_gpu.py`` holds the production settings against.

    python tests/golden/make_converged_spectra.py --lx1 6 --engine direct      # numpy oracle, sparse direct solves
    python tests/golden/make_converged_spectra.py --lx1 8 --engine port        # C + OpenMP port, iterative solves at 1e-13 / 1e-9

Both run the reference's sequence (core/eigensolvers.f:216-282, core/krylov_decomposition.f:73-202): seed = add_noise,
normalise, one matvec, normalise, then k_dim = 200 Arnoldi steps with two-pass Gram-Schmidt in the bm1s inner product,
``eig`` of the Hessenberg matrix, residual |H(k+1,k) y_k| (core/eigensolvers.f:346-350).  Output: rows
[Re mu, Im mu, residual] sorted by decreasing modulus, merged into tests/golden/cylinder_oracle_spectra.npz under
``Hd<lx1>`` together with the run's provenance (engine, tolerances, wall time).  Run once in the build container
(lx1 = 6 direct: ~1 h on one core; lx1 = 8 port: ~1.5 h on four cores); the Hessenberg matrix is checkpointed to
/tmp every 10 steps.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from nekstab_amd import mesh, seed  # noqa: E402
from oracle.linns import LinNS2D, ritz  # noqa: E402

OUT = os.path.join(HERE, "cylinder_oracle_spectra.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lx1", type=int, required=True)
    ap.add_argument("--engine", choices=("direct", "port"), required=True)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--tol-helm", type=float, default=1e-13)
    ap.add_argument("--tol-pres", type=float, default=1e-9)
    ap.add_argument("--adjoint", action="store_true")
    ap.add_argument("--key", default=None)
    a = ap.parse_args()
    case = mesh.load_case_npz(os.path.join(HERE, "cylinder_case.npz"), a.lx1, adjoint=a.adjoint)
    t0 = time.time()
    o = LinNS2D(x=case.x, y=case.y, gid=case.gid, nglob=case.nglob, mask=case.mask, ub=case.ub, spng=case.spng, re=case.re,
                endtime=case.endtime, lxd=case.lxd, has_outflow=case.has_outflow, factorize_pressure=(a.engine == "direct"))
    if a.engine == "port":
        assert not a.adjoint
        from oracle.cpu_port import CpuPort
        eng = CpuPort(o, case.meta["vert"], case.meta["nvert"], tol_helm=a.tol_helm, tol_pres=a.tol_pres, tol_relative=1,
                      max_helm=400, max_pres=48, early_pres_mul=1.0)
        eng.set_threads(a.threads)
        matvec = eng.matvec
    else:
        matvec = lambda q: o.matvec(q, a.adjoint)
    print("set-up %.1f s, nsteps %d" % (time.time() - t0, o.nsteps), flush=True)
    w = o.bm1s()
    qx, qy = seed.add_noise(case)
    q = (qx, qy, np.zeros((o.nel, o.m, o.m)))
    nrm = np.sqrt(o.inner(q, q, w))
    q = tuple(c / nrm for c in q)
    q = matvec(q)                                       # core/eigensolvers.f:234: the seed goes through the map once
    nrm = np.sqrt(o.inner(q, q, w))
    Q = [tuple(c / nrm for c in q)]
    k = a.k
    H = np.zeros((k + 1, k))
    key = a.key or ("H%s%d" % ("a" if a.adjoint else "d", a.lx1))
    t0 = time.time()
    for j in range(k):
        f = list(matvec(Q[j]))
        for _ in range(2):
            for i in range(j + 1):
                al = o.inner(f, Q[i], w)
                f = [fc - al * qc for fc, qc in zip(f, Q[i])]
                H[i, j] += al
        beta = np.sqrt(o.inner(f, f, w))
        H[j + 1, j] = beta
        Q.append(tuple(c / beta for c in f))
        if (j + 1) % 10 == 0 or j + 1 == k:
            np.save("/tmp/%s_H.npy" % key, H)
            vals, _, res = ritz(H, j + 1)
            print("step %3d  %.0f s  leading %.10f%+.10fi  residual %.1e" % (j + 1, time.time() - t0, vals[0].real, abs(vals[0].imag), res[0]), flush=True)
    vals, _, res = ritz(H, k)
    tab = np.stack([vals.real, vals.imag, res], axis=1)
    old = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    old[key] = tab
    old[key + "_hess"] = H
    prov = "engine=%s lx1=%d k=%d adjoint=%d" % (a.engine, a.lx1, k, a.adjoint)
    if a.engine == "port":
        prov += " tol_helm=%g tol_pres=%g relative threads=%d" % (a.tol_helm, a.tol_pres, a.threads)
    else:
        prov += " sparse direct solves (SuperLU)"
    old[key + "_provenance"] = np.array(prov + " wall=%.0fs" % (time.time() - t0))
    np.savez_compressed(OUT, **old)
    print("wrote", OUT, key, prov)
    for r in tab[:24]:
        print("  %.10f %+.10f  %.1e" % tuple(r))


if __name__ == "__main__":
    main()
