"""Newton-Krylov fixed-point solver on the time-stepper map, restating
    newton_krylov              core/newton_krylov.f:5-167
    ts_gmres                   core/newton_krylov.f:175-296
    initialize_gmres_vector    core/newton_krylov.f:305-328
    nonlinear_forward_map      core/newton_krylov.f:336-378   (-> nsk_nonlinear_map on the device)
    newton_linearized_map      core/matvec.f:381-428          (-> nsk_matvec mode NSK_NEWTON)
over the device backend: steady fixed points (uparam(1) = 2) and unstable periodic orbits (uparam(1) = 2.1: the Krylov
vector carries the period as its ``time`` component, core/krylov_subspace.f:12, :46-48, and the linearised map is bordered
with the time derivatives at the two ends of the orbit, core/matvec.f:402-418, :434-475)."""
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


# ----------------------------------------------------------------------------------------------------------------------
# Newton for unstable periodic orbits (uparam(1) = 2.1)
# ----------------------------------------------------------------------------------------------------------------------
class BVec:
    """krylov_vector with its ``time`` component (core/krylov_subspace.f:7-15)."""
    def __init__(self, v, t=0.0):
        self.v, self.t = v, float(t)


class BorderedBackend:
    """The vector interface of krylov.py / ts_gmres on bordered vectors: velocity-pressure state on the device + the period
    on the host.  ``matvec`` is newton_linearized_map's UPO branch (core/matvec.f:398-418):
        f      = (exp(LT) - I) q + bvec(fc_nwt) * q%time
        f%time = < bvec(ic_nwt), q >
    with bvec(state) = (one full-equation time step from state - state) / dt (compute_bvec, core/matvec.f:434-475)."""

    def __init__(self, be):
        self.be = be
        self.bvec, self.btvec = be.alloc(2)

    def close(self):
        self.be.free([self.bvec, self.btvec])

    def time_derivative(self, out, state):
        """compute_bvec: a single first-order step of the full equations from ``state`` approximates d(state)/dt."""
        be = self.be
        ns = be.nsteps
        be.set_nsteps(1)
        be.nonlinear_map(out, state, subtract_q=True)
        be.set_nsteps(ns)
        be.scal(out, 1.0 / be.dt)

    # ---- vector interface
    def alloc(self, n=1):
        return [BVec(v) for v in self.be.alloc(n)]

    def free(self, vs):
        self.be.free([b.v for b in vs])

    def copy(self, dst, src):
        self.be.copy(dst.v, src.v); dst.t = src.t

    def zero(self, p):
        self.be.zero(p.v); p.t = 0.0

    def scal(self, p, a):
        self.be.scal(p.v, a); p.t *= a

    def axpy(self, p, a, q):
        self.be.axpy(p.v, a, q.v); p.t += a * q.t

    def dot(self, p, q):
        return self.be.dot(p.v, q.v) + p.t * q.t             # core/krylov_subspace.f:46-48

    def norm(self, p):
        return float(np.sqrt(self.dot(p, p)))

    def orth(self, f, Q):
        h = np.zeros(len(Q))
        for _ in range(2):                                   # update_hessenberg_matrix: two passes
            for i, q in enumerate(Q):
                c = self.dot(f, q)
                self.axpy(f, -c, q)
                h[i] += c
        beta = self.norm(f)
        self.scal(f, 1.0 / beta)
        return h, beta

    def basis_gemv(self, Q, y, out, im=None):
        self.be.basis_gemv([q.v for q in Q], y, out.v)
        out.t = float(sum(np.real(yi) * q.t for yi, q in zip(y, Q)))

    def matvec(self, f, q, mode=NSK_NEWTON):
        be = self.be
        be.matvec(f.v, q.v, NSK_NEWTON)                      # (exp(LT) - I) q along the stored orbit
        be.axpy(f.v, q.t, self.bvec)                         # + d/dt at the end of the orbit * period correction
        f.t = be.dot(self.btvec, q.v)                        # phase condition: no shift along the orbit at t = 0


def newton_krylov_upo(be, q, period, k_dim=100, tol=1e-11, maxiter_newton=20, spng_str=0.0, log=None):
    """q <- a point of a periodic orbit of the full equations, period <- its period (core/newton_krylov.f:5-167 with
    uparam(1) = 2.1).  ``q``: device vector holding the initial guess; returns (period, iterations, residual history)."""
    bb = BorderedBackend(be)
    fc = be.alloc(1)[0]
    f, dq = bb.alloc(2)
    qb = BVec(q, period)
    hist = []
    for i in range(1, maxiter_newton + 1):
        be.set_option("endtime", qb.t)                       # param(10) = q%time; prepare_linearized_solver
        be.set_orbit(qb.v, spng_str=spng_str, end=fc)        # nonlinear_forward_map, the orbit stored for the linearised maps
        be.copy(f.v, fc)
        be.axpy(f.v, -1.0, qb.v)
        f.t = 0.0
        residual = bb.norm(f) ** 2
        hist.append((residual, qb.t))
        if log:
            log("newton", i, residual, qb.t)
        if residual < tol:
            break
        bb.time_derivative(bb.bvec, fc)                      # compute_bvec(bvec, fc_nwt)
        bb.time_derivative(bb.btvec, qb.v)                   # compute_bvec(btvec, ic_nwt)
        ts_gmres(bb, f, dq, k_dim, tol, log=log)
        bb.axpy(qb, -1.0, dq)                                # krylov_sub2(q, dq), period included
    bb.free([f, dq]); be.free([fc]); bb.close()
    return qb.t, i, hist
