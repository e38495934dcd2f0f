"""Arnoldi checkpoint / restart in the reference's file formats
(``arnoldi_checkpoint`` core/eigensolvers.f:802-905, restart branch of ``krylov_schur`` :284-325,
``load_files`` core/IO.f:15-60):

  KRY<session>0.f<k+1>        Krylov vector k+1 as a Nek field file (pressure on mesh 1)
  HES<session><k:04d>         H(1:k+1, 1:k), list-directed, row by row
  Spectre_H<op><k:04d>.dat    intermediate spectra, (3E15.7)
  Spectre_NS<op><k:04d>.dat

The reference writes KRY files in single precision (writeDoublePrecision = no); ``wdsize=8`` is the
default here so that a restarted factorisation continues the same Krylov sequence to round-off.
"""
from __future__ import annotations

import os

import numpy as np

from . import nekio
from .krylov import eig_sorted, log_transform
from .quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix


def _maps(case):
    n, m = case.lx1, case.lx1 - 2
    zl, zg = gauss_lobatto_legendre(n)[0], gauss_legendre(m)[0]
    return interp_matrix(zg, zl), interp_matrix(zl, zg)          # map21 (output), map12 (input)


def kry_name(session, i):
    return "KRY%s0.f%05d" % (session, i)


def state_to_fields(be, case, v):
    """Device state -> (x, u, p on mesh 1) in the field-file layout (ndim, nel, nz, ny, nx); 2-D and 3-D."""
    J21, _ = _maps(case)
    if getattr(case, "ndim", 2) == 3:
        vx, vy, vz, pr = be.download3(v)
        p1 = np.einsum("ci,bj,ak,ekji->ecba", J21, J21, J21, pr, optimize=True)      # map21 along r, s, t
        return np.stack([case.x, case.y, case.z]), np.stack([vx, vy, vz]), p1
    vx, vy, pr = be.download(v)
    return np.stack([case.x, case.y])[:, :, None], np.stack([vx, vy])[:, :, None], (J21 @ pr @ J21.T)[:, None]


def fields_to_state(be, case, v, f):
    """Field file (``nekio.NekField``) -> device state (pressure back to mesh 2: map12)."""
    _, J12 = _maps(case)
    if getattr(case, "ndim", 2) == 3:
        p2 = np.einsum("ci,bj,ak,ekji->ecba", J12, J12, J12, f.p, optimize=True)
        be.upload3(v, f.u[0], f.u[1], f.u[2], p2)
    else:
        be.upload(v, f.u[0, :, 0], f.u[1, :, 0], J12 @ f.p[:, 0] @ J12.T)


def write_krylov_vector(be, case, v, path, time=0.0, wdsize=8):
    x, u, p1 = state_to_fields(be, case, v)
    nekio.write_fld(path, x=x, u=u, p=p1, time=time, istep=be.nsteps + 1, wdsize=wdsize)


def read_krylov_vector(be, case, v, path):
    fields_to_state(be, case, v, nekio.read_fld(path))


def arnoldi_checkpoint(be, case, Q, H, k, outdir, *, session="1cyl", evop="d", sampling_period=1.0, wdsize=8):
    """After Arnoldi step k (1-based): vector k+1, the Hessenberg matrix and the current spectra."""
    os.makedirs(outdir, exist_ok=True)
    if k == 1:
        write_krylov_vector(be, case, Q[0], os.path.join(outdir, kry_name(session, 1)), 0.0, wdsize)   # :280-282
    write_krylov_vector(be, case, Q[k], os.path.join(outdir, kry_name(session, k + 1)), float(k), wdsize)
    vals, vecs = eig_sorted(H[:k, :k])
    res = np.abs(H[k, k - 1] * vecs[k - 1, :])
    nekio.write_spectre(os.path.join(outdir, "Spectre_H%s%04d.dat" % (evop, k)), vals, res)
    nekio.write_spectre(os.path.join(outdir, "Spectre_NS%s%04d.dat" % (evop, k)), log_transform(vals, sampling_period), res)
    with open(os.path.join(outdir, "HES%s%04d" % (session, k)), "w") as f:
        f.write(" ".join("%.17g" % x for x in H[:k + 1, :k].ravel()) + "\n")          # row-major, as `write(67,*)`


def load_checkpoint(be, case, indir, k_dim, mstart, *, session="1cyl"):
    """Restart state for ``uparam(2) = mstart``: H from HES<session><mstart>, Q(1..mstart+1) from the KRY files.
    Returns (Q, H, next_mstart) ready for ``arnoldi_factorization(be, Q, H, next_mstart, k_dim)``."""
    vals = np.array(open(os.path.join(indir, "HES%s%04d" % (session, mstart))).read().split(), dtype=float)
    Hm = vals[: (mstart + 1) * mstart].reshape(mstart + 1, mstart)
    H = np.zeros((k_dim + 1, k_dim))
    # k_dim < mstart ("subsampling", core/eigensolvers.f:295-301): the leading (k_dim+1) x k_dim block of the checkpointed matrix
    # with the first k_dim+1 Krylov vectors -- a leading part of an Arnoldi factorisation is itself one, so the run goes on
    # with the eigen-decomposition / Krylov-Schur restart of a FULL factorisation (next_mstart = k_dim + 1: no Arnoldi step left).
    # (The reference reads that block with one formatted READ per entry from a file it wrote list-directed, and then loads
    # mstart vectors into k_dim+1 slots: its branch cannot run as written; this is what it is for.)
    m = min(mstart, k_dim)
    H[: m + 1, :m] = Hm[: m + 1, :m]
    Q = be.alloc(k_dim + 1)
    for i in range(1, m + 2):
        read_krylov_vector(be, case, Q[i - 1], os.path.join(indir, kry_name(session, i)))
    return Q, H, m + 1
