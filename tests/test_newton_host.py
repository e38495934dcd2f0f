"""Host Newton-Krylov logic (nekstab_amd/newton.py: newton_krylov, ts_gmres -- core/newton_krylov.f:5-296) on a dense numpy
backend with a toy nonlinear 'time-stepper' map: no GPU, no oracle; the GPU runs of the same code against the reference's base
flows are in tests/test_newton_gpu.py."""
import numpy as np

from nekstab_amd import newton
from nekstab_amd.capi import NSK_NEWTON
from tests.dense_backend import DenseBackend


class ToyMap(DenseBackend):
    """Phi(q) = M q + c + eps q*q (componentwise); the NEWTON matvec is (DPhi(base) - I), as core/matvec.f:381-401."""

    def __init__(self, n=40, eps=0.02, seed=5):
        rng = np.random.default_rng(seed)
        M = 0.3 * rng.standard_normal((n, n)) / np.sqrt(n)          # contraction: a real fixed point exists near (I - M)^-1 c
        super().__init__(M, w=1.0 + rng.random(n))
        self.c = rng.standard_normal(n)
        self.eps = eps
        self.base = np.zeros(n)
        self.nset = 0

    def phi(self, x):
        return self.A @ x + self.c + self.eps * x * x

    def free(self, vs):
        pass

    def zero(self, p):
        p.a = np.zeros(self.n)

    def axpy(self, p, a, q):
        p.a = p.a + a * q.a

    def set_baseflow(self, q):
        self.base = q.a.copy()
        self.nset += 1

    def nonlinear_map(self, f, q, subtract_q=False):
        f.a = self.phi(q.a) - (q.a if subtract_q else 0.0)

    def matvec(self, f, q, mode=0):
        assert mode == NSK_NEWTON
        f.a = self.A @ q.a + 2.0 * self.eps * self.base * q.a - q.a
        self.nmat += 1


def test_ts_gmres_solves_the_newton_system():
    be = ToyMap()
    be.base = np.linspace(-1, 1, be.n)
    J = be.A + np.diag(2.0 * be.eps * be.base) - np.eye(be.n)
    rng = np.random.default_rng(1)
    rhs, sol = be.alloc(2)
    rhs.a = rng.standard_normal(be.n)
    log = []
    calls = newton.ts_gmres(be, rhs, sol, k_dim=be.n, tol=1e-24, log=lambda *a: log.append(a))
    assert np.abs(J @ sol.a - rhs.a).max() < 1e-10
    assert calls == be.nmat and calls <= be.n + 2
    res = [r for tag, k, r in log if tag == "arnoldi"]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(res, res[1:]))              # GMRES residuals do not increase
    # restarted: a Krylov space of 8 needs several cycles and still converges
    be2 = ToyMap()
    be2.base = be.base.copy()
    rhs2, sol2 = be2.alloc(2)
    rhs2.a = rhs.a.copy()
    log2 = []
    newton.ts_gmres(be2, rhs2, sol2, k_dim=8, tol=1e-22, log=lambda *a: log2.append(a))
    assert sum(1 for tag, *_ in log2 if tag == "gmres") >= 2
    assert np.abs(J @ sol2.a - rhs2.a).max() < 1e-9


def test_newton_krylov_reaches_the_fixed_point_quadratically():
    be = ToyMap()
    q = be.alloc(1)[0]
    q.a = np.zeros(be.n)
    it, hist = newton.newton_krylov(be, q, k_dim=be.n, tol=1e-26, maxiter_newton=20)
    assert np.abs(be.phi(q.a) - q.a).max() < 1e-12
    assert it <= 8 and be.nset == it
    # quadratic convergence of |f|^2 (the history holds squared residuals): r_{k+1} <~ C r_k^2 once converging
    h = [x for x in hist if x > 0]
    k = next(i for i, x in enumerate(h) if x < 1e-3)
    assert h[k + 1] < 10.0 * h[k] ** 2 + 1e-28
