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


class OrbitBackend:
    """A planar oscillator with a limit cycle (x' = mu x - w y - r^2 x, y' = w x + mu y - r^2 y: radius sqrt(mu), period
    2 pi / w) behind the interface newton_krylov_upo uses: full-equation map over `nsteps` RK4 steps of dt = endtime / nsteps,
    the orbit stored by set_orbit, the NEWTON matvec = monodromy along that orbit minus the identity."""

    def __init__(self, mu=0.25, w=1.3, nsteps=200):
        self.mu, self.w = mu, w
        self.nsteps_full, self.nsteps = nsteps, nsteps
        self.endtime = 1.0
        self.orbit = None

    @property
    def dt(self):
        return self.endtime / self.nsteps_full

    def rhs(self, z):
        x, y = z
        r2 = x * x + y * y
        return np.array([self.mu * x - self.w * y - r2 * x, self.w * x + self.mu * y - r2 * y])

    def jac(self, z):
        x, y = z
        r2 = x * x + y * y
        return np.array([[self.mu - r2 - 2 * x * x, -self.w - 2 * x * y], [self.w - 2 * x * y, self.mu - r2 - 2 * y * y]])

    def rk4(self, z, h):
        k1 = self.rhs(z); k2 = self.rhs(z + 0.5 * h * k1); k3 = self.rhs(z + 0.5 * h * k2); k4 = self.rhs(z + h * k3)
        return z + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    # ---- vector interface
    def alloc(self, n=1):
        return [DenseVec() for _ in range(n)]

    def free(self, vs):
        pass

    def copy(self, dst, src):
        dst.a = src.a.copy()

    def zero(self, p):
        p.a = np.zeros(2)

    def scal(self, p, a):
        p.a = p.a * a

    def axpy(self, p, a, q):
        p.a = p.a + a * q.a

    def dot(self, p, q):
        return float(p.a @ q.a)

    def basis_gemv(self, Q, y, out, im=None):
        out.a = sum(np.real(yi) * q.a for yi, q in zip(y, Q))

    # ---- time steppers
    def set_option(self, name, value):
        assert name == "endtime"
        self.endtime = float(value)

    def set_nsteps(self, n):
        self.nsteps = n

    def nonlinear_map(self, out, state, subtract_q=False):
        z = state.a.copy()
        for _ in range(self.nsteps):
            z = self.rk4(z, self.dt)
        out.a = z - (state.a if subtract_q else 0.0)

    def set_orbit(self, q, spng_str=0.0, end=None):
        z = q.a.copy()
        self.orbit = [z.copy()]
        for _ in range(self.nsteps_full):
            z = self.rk4(z, self.dt)
            self.orbit.append(z.copy())
        if end is not None:
            end.a = z.copy()

    def matvec(self, f, q, mode=0):
        assert mode == NSK_NEWTON and self.orbit is not None
        v, h = q.a.copy(), self.dt
        for z in self.orbit[:-1]:                              # RK4 on the variational equation, Jacobian frozen per step
            J = self.jac(z)
            k1 = J @ v; k2 = J @ (v + 0.5 * h * k1); k3 = J @ (v + 0.5 * h * k2); k4 = J @ (v + h * k3)
            v = v + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        f.a = v - q.a


class DenseVec:
    def __init__(self):
        self.a = np.zeros(2)


def test_newton_for_periodic_orbits_finds_the_limit_cycle():
    """newton_krylov_upo + BorderedBackend (core/newton_krylov.f with uparam(1) = 2.1, core/matvec.f:402-475): from a point
    off the cycle and a period 8 % off, Newton on the bordered system returns a point ON the cycle and its period."""
    be = OrbitBackend()
    q = be.alloc(1)[0]
    q.a = np.array([0.58, 0.07])                                # radius 0.584 instead of 0.5
    T0 = 0.92 * 2.0 * np.pi / be.w
    period, it, hist = newton.newton_krylov_upo(be, q, T0, k_dim=3, tol=1e-22, maxiter_newton=25)
    assert it < 25 and hist[-1][0] < 1e-22
    assert abs(period - 2.0 * np.pi / be.w) < 1e-6              # RK4 with 200 steps per period: O(dt^4)
    assert abs(np.hypot(*q.a) - np.sqrt(be.mu)) < 1e-6
    # the returned state closes the orbit under the full map
    out = be.alloc(1)[0]
    be.set_option("endtime", period)
    be.nonlinear_map(out, q, subtract_q=True)
    assert np.abs(out.a).max() < 1e-10
