"""Newton-Krylov fixed-point solver on the time-stepper map, restating
    newton_krylov              core/newton_krylov.f:5-167
    ts_gmres                   core/newton_krylov.f:175-296
    initialize_gmres_vector    core/newton_krylov.f:305-328
    nonlinear_forward_map      core/newton_krylov.f:336-378   (-> nsk_nonlinear_map on the device)
    newton_linearized_map      core/matvec.f:381-428          (-> nsk_matvec mode NSK_NEWTON)
over the device backend (steady fixed points, uparam(1) = 2; the UPO variants are not built)."""
from __future__ import annotations

import numpy as np

from .capi import NSK_NEWTON
from . import krylov


def ts_gmres(be, rhs, sol, k_dim, tol, maxiter=100, log=None):
    """GMRES on (exp(LT) - I) with the reference's restart logic; returns matvec count."""
    Q = be.alloc(k_dim + 1)
    dq, f = be.alloc(2)
    be.zero(sol)
    be.copy(Q[0], rhs)
    beta = be.norm(Q[0])
    be.scal(Q[0], 1.0 / beta)
    calls = 0
    for it in range(1, maxiter + 1):
        H = np.zeros((k_dim + 1, k_dim))
        evec = np.zeros(k_dim + 1)
        evec[0] = beta
        k_used = k_dim
        for k in range(1, k_dim + 1):
            krylov.arnoldi_factorization(be, Q, H, k, k, NSK_NEWTON)            # one column at a time (:255)
            y = np.linalg.lstsq(H[:k + 1, :k], evec[:k + 1], rcond=None)[0]      # lstsq -> dgels (:258)
            res = np.linalg.norm(evec[:k + 1] - H[:k + 1, :k] @ y)
            calls += 1
            if log:
                log("arnoldi", k, res ** 2)
            if res ** 2 < tol:
                k_used = k
                break
        be.basis_gemv(Q[:k_used], y[:k_used], dq)                               # krylov_matmul (:275)
        be.axpy(sol, 1.0, dq)
        # sanity residual and new seed: f = rhs - A sol  (initialize_gmres_vector)
        be.matvec(f, sol, NSK_NEWTON)
        calls += 1
        be.axpy(f, -1.0, rhs)
        be.scal(f, -1.0)
        beta = be.norm(f)
        if log:
            log("gmres", it, beta ** 2)
        if beta ** 2 < tol:
            break
        be.copy(Q[0], f)
        be.scal(Q[0], 1.0 / beta)
    be.free(Q + [dq, f])
    return calls


def newton_krylov(be, q, k_dim=100, tol=1e-11, maxiter_newton=100, log=None):
    """q <- fixed point of the nonlinear map Phi_T.  Returns (iterations, residual history)."""
    f, dq = be.alloc(2)
    hist = []
    for i in range(1, maxiter_newton + 1):
        be.set_baseflow(q)                               # prepare_linearized_solver on the current iterate
        be.nonlinear_map(f, q, subtract_q=True)          # f(q) = Phi_T(q) - q
        residual = be.norm(f) ** 2
        hist.append(residual)
        if log:
            log("newton", i, residual)
        if residual < tol:
            break
        ts_gmres(be, f, dq, k_dim, tol, log=log)
        be.axpy(q, -1.0, dq)                             # krylov_sub2(q, dq)
    be.free([f, dq])
    return i, hist
