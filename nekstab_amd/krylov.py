"""Host side of the eigensolver: Arnoldi factorisation and Krylov-Schur restart,
mirroring the reference's host routines call for call, over the device operator.

    arnoldi_factorization      core/krylov_decomposition.f:7-104
    update_hessenberg_matrix   core/krylov_decomposition.f:116-202  (-> nsk_orth on the device)
    krylov_schur               core/eigensolvers.f:141-388
    schur_condensation         core/eigensolvers.f:395-499
    select_eigenvalues         core/eigensolvers.f:729-795
    eig / schur / ordschur     core/lapack_wrapper.f:7-251 (dgeev, dgees, dtrsen via SciPy's LAPACK)
    outpost_ks / log_transform core/eigensolvers.f:508-721, :908-915

The Krylov basis lives on the device (opaque ``nsk_vec`` handles); only the
(k+1) x k Hessenberg matrix and k x k dense factorisations are on the host,
exactly as in the reference where they are replicated on every MPI rank.

``backend`` is anything with the NekStabHip vector interface (alloc/matvec/
orth/norm/scal/copy/basis_gemm/basis_gemv); tests drive the same code with a
small dense numpy backend to check the restart logic without a GPU.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg as sla


@dataclass
class KrylovResult:
    vals: np.ndarray            # Ritz values mu (Hessenberg spectrum), sorted by decreasing |mu|
    vecs: np.ndarray            # eigenvectors of H(1:k,1:k)
    residual: np.ndarray        # |H(k+1,k) * y_k|
    H: np.ndarray
    Q: list                     # device handles (k+1)
    matvecs: int = 0
    schur_cnt: int = 0
    wall: float = 0.0
    timings: dict = field(default_factory=dict)


def eig_sorted(Hk: np.ndarray):
    """``eig`` of core/lapack_wrapper.f:129-251: dgeev + sort by decreasing modulus."""
    vals, vecs = sla.eig(Hk)
    order = np.argsort(-np.abs(vals), kind="stable")
    return vals[order], vecs[:, order]


def select_eigenvalues(vals: np.ndarray, schur_del: float, schur_tgt: int):
    """core/eigensolvers.f:729-795: everything outside the circle 1-delta, at least the
    nev+4 largest, plus the conjugate partner of the smallest selected one."""
    n = len(vals)
    idx = np.argsort(np.abs(vals), kind="stable")            # ascending, like quicksort2
    selected = np.abs(vals) >= (1.0 - schur_del)
    lo = max(0, n - (schur_tgt + 4))
    selected[idx[lo:]] = True
    if lo - 1 >= 0 and vals[idx[lo]].imag == -vals[idx[lo - 1]].imag:
        selected[idx[lo - 1]] = True
    return selected


def _schur_block_eigs(T):
    """(wr, wi) in diagonal-block order, as dgees returns them."""
    k = T.shape[0]
    out = np.zeros(k, dtype=complex)
    i = 0
    while i < k:
        if i + 1 < k and T[i + 1, i] != 0.0:
            w = np.linalg.eigvals(T[i:i + 2, i:i + 2])
            w = w[np.argsort(-w.imag)]                        # LAPACK: positive imaginary part first
            out[i], out[i + 1] = w[0], w[1]
            i += 2
        else:
            out[i] = T[i, i]
            i += 1
    return out


def arnoldi_factorization(be, Q, H, mstart, mend, mode=0, log=None, stats=None):
    """k-step Arnoldi  M Q_k = Q_{k+1} H  (core/krylov_decomposition.f:73-102)."""
    for mstep in range(mstart, mend + 1):             # 1-based like the reference
        t0 = time.perf_counter()
        f = Q[mstep]                                  # slot of the new vector (0-based: Q[mstep] is vector mstep+1)
        be.matvec(f, Q[mstep - 1], mode)
        t1 = time.perf_counter()
        h, beta = be.orth(f, Q[:mstep])
        H[:mstep, mstep - 1] = h
        H[mstep, mstep - 1] = beta
        t2 = time.perf_counter()
        if stats is not None:
            stats.setdefault("matvec_s", []).append(t1 - t0)
            stats.setdefault("orth_s", []).append(t2 - t1)
        if log:
            log(mstep, H, t2 - t0)
    return H


def schur_condensation(be, Q, H, k, mstart, schur_del, schur_tgt):
    """core/eigensolvers.f:395-499. Returns the new ``mstart``."""
    from scipy.linalg import lapack
    b = np.zeros(k)
    b[k - 1] = H[k, k - 1]
    # schur -> dgees('V','S', |lambda| > 0.9)            core/lapack_wrapper.f:7-59, :258-270
    T, Z, _ = sla.schur(H[:k, :k], output="real", sort=lambda re, im: np.hypot(re, im) > 0.9)
    vals = _schur_block_eigs(T)
    sel = select_eigenvalues(vals, schur_del, schur_tgt)
    ms = int(sel.sum())
    # ordschur -> dtrsen('N','V')                         core/lapack_wrapper.f:70-122
    out = lapack.dtrsen(sel.astype(np.int32), np.asfortranarray(T), np.asfortranarray(Z), job="N", wantq=1)
    T2, Z2, info = out[0], out[1], out[-1]
    if info != 0:
        raise RuntimeError(f"dtrsen info={info}")
    Hn = np.zeros_like(H)
    Hn[:ms, :ms] = T2[:ms, :ms]                    # :451-452 zero the unwanted blocks
    be.basis_gemm(Q[:k], Z2)                       # Q(:,1:k) <- Q(:,1:k) Z   (:455-474)
    Hn[ms, :ms] = (b @ Z2)[:ms]                     # b^T Z                     (:478-479)
    be.copy(Q[ms], Q[k])                           # last Krylov vector restarts the factorisation
    H[:, :] = Hn
    return ms + 1


def krylov_schur(be, q0, k_dim, *, mode=0, schur_tgt=0, eigen_tol=1e-6, schur_del=0.10,
                 max_restarts=50, log=None):
    """Krylov-Schur eigensolver (core/eigensolvers.f:141-388). ``q0`` is a device
    vector already holding the (un-normalised) seed; ``schur_tgt <= 0`` = plain
    k-step Arnoldi (the committed cylinder example, 1cyl.usr:22)."""
    t_start = time.perf_counter()
    Q = be.alloc(k_dim + 1)
    H = np.zeros((k_dim + 1, k_dim))
    be.copy(Q[0], q0)
    nrm = be.norm(Q[0])
    be.scal(Q[0], 1.0 / nrm)                       # krylov_normalize, :271-278
    mstart, schur_cnt, nmat = 1, 0, 0
    stats = {}
    while True:
        arnoldi_factorization(be, Q, H, mstart, k_dim, mode, log, stats)
        nmat += k_dim - mstart + 1
        vals, vecs = eig_sorted(H[:k_dim, :k_dim])
        residual = np.abs(H[k_dim, k_dim - 1] * vecs[k_dim - 1, :])      # :349
        cnt = int(np.sum(residual < eigen_tol))
        if schur_tgt <= 0 or cnt >= schur_tgt or schur_cnt >= max_restarts:
            break
        schur_cnt += 1
        mstart = schur_condensation(be, Q, H, k_dim, mstart, schur_del, schur_tgt)
    return KrylovResult(vals, vecs, residual, H, Q, nmat, schur_cnt,
                        time.perf_counter() - t_start, stats)


def log_transform(vals: np.ndarray, T: float) -> np.ndarray:
    """lambda = log(mu)/T  (core/eigensolvers.f:908-915)."""
    return np.log(vals.astype(complex)) / T


def assemble_mode(be, res: KrylovResult, i: int, re, im):
    """Eigenmode i = Q y_i, normalised so that |Re|^2+|Im|^2 = 1 in the bm1s norm
    (core/eigensolvers.f:607-627)."""
    k = res.H.shape[1]
    be.basis_gemv(res.Q[:k], res.vecs[:, i], re, im)
    a = be.dot(re, re) + be.dot(im, im)
    s = 1.0 / np.sqrt(a)
    be.scal(re, s)
    be.scal(im, s)


def band_arnoldi(be, seeds, k_dim, *, mode=0, log=None):
    """Band (block) Arnoldi in Ruhe's form with band width b = len(seeds):  M Q_k = Q_{k+b} H,  H banded upper Hessenberg with
    b sub-diagonals.  Column i is produced by orthogonalising M q_i against ALL vectors made so far (the reference's
    update_hessenberg_matrix, one vector at a time), so the b maps M q_i .. M q_{i+b-1} are independent and run as ONE
    ``matvec_batch`` -- b maps in flight on b lanes of the GPU (nsk_matvec_batch): 1.6x the matvecs/s of the single-vector
    factorisation at b = 2 on BASELINE config 2.  The price is the polynomial degree: a space of k_dim vectors holds degree
    k_dim / b per seed, so a dominant pair converges in about as many BATCHES as the single-vector run needs steps; the band
    form pays when several eigenpairs (or several seeds: direct sweeps) are wanted.  Not in the reference (one MPI job = one
    map); the single-vector ``krylov_schur`` stays the pinned default.

    ``seeds``: b device vectors (any, linearly independent).  Returns a KrylovResult with H of shape (k_dim + b, k_dim),
    the Ritz pairs of H[:k_dim, :k_dim] and residuals |H[k_dim:k_dim+b, :k_dim] y| (core/eigensolvers.f:346-350 generalised)."""
    b = len(seeds)
    t_start = time.perf_counter()
    Q = be.alloc(k_dim + b)
    H = np.zeros((k_dim + b, k_dim))
    stats = {}
    for j in range(b):                                   # orthonormal seeds
        be.copy(Q[j], seeds[j])
        be.orth(Q[j], Q[:j])
    i = 0
    while i < k_dim:
        nb = min(b, k_dim - i)
        t0 = time.perf_counter()
        fs = [Q[i + b + j] for j in range(nb)]
        if nb > 1:
            be.matvec_batch(fs, [Q[i + j] for j in range(nb)], mode)
        else:
            be.matvec(fs[0], Q[i], mode)
        t1 = time.perf_counter()
        for j in range(nb):
            n = i + b + j                                # vectors made so far
            h, beta = be.orth(fs[j], Q[:n])
            H[:n, i + j] = h
            H[n, i + j] = beta
        t2 = time.perf_counter()
        stats.setdefault("matvec_s", []).append(t1 - t0)
        stats.setdefault("orth_s", []).append(t2 - t1)
        if log:
            log(i + nb, H, t2 - t0)
        i += nb
    vals, vecs = eig_sorted(H[:k_dim, :k_dim])
    residual = np.linalg.norm(H[k_dim:k_dim + b, :k_dim] @ vecs, axis=0)
    return KrylovResult(vals, vecs, residual, H, Q, k_dim, 0, time.perf_counter() - t_start, stats)
