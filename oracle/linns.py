"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module. The product path (``nekstab_amd`` + the HIP C-ABI
library) never calls it and fails loudly when the HIP library is missing.

What it restates (numpy/scipy, fp64):

  * nekStab's time-stepper matvec  ``f = exp(L T) q``
      core/matvec.f:1-52    (dt / nsteps rule, ``prepare_linearized_solver``)
      core/matvec.f:163-243 (``forward_linearized_map``: load q, nsteps x nek_advance, read f)
      core/matvec.f:249-326 (``adjoint_linearized_map``)
      core/utils.f:149-180  (``nekStab_forcing`` perturbation branch: ff -= spng_fun*u')
  * the Krylov vector algebra
      core/krylov_subspace.f:24-56  (bm1s-weighted inner product, pressure excluded)
      core/krylov_decomposition.f:116-202 (two-pass modified Gram-Schmidt)
  * what ``nek_advance`` does in perturbation mode.  Nek5000 itself is an
    un-vendored, un-pinned dependency (Nek5000clone.sh:3-7, fork nekStab/Nek5000
    @master), so that part restates the *published* Nek5000 v19 algorithm
    (perturb.f: fluidp/perturbv/makefp/advabp/advabp_adjoint/makextp/makebdfp/
    cresvipp/incomprp/extrapprp; navier1.f: opgradt/opdiv/cdabdtp/opbinv;
    hmholtz.f: axhelm; convect.f: convect_new/convect_adj; subs1.f: setordbd,
    compute_cfl) -- SURVEY.md Appendix A.

Parity pins (tests/test_oracle_golden.py): nsteps = 100 @lx1=6 / 183 @lx1=8 (field
headers), mass-matrix sum, mode normalisation under the bm1s inner product, the
eigen-relation  M(dRe+i dIm) = mu (dRe + i dIm) with mu from Spectre_Hd.dat and
the leading Ritz pair of a short Arnoldi run (tests/golden/).

Everything *iterative* in Nek (PCG, GMRES, Schwarz) is replaced here by sparse
direct solves: only the converged solution is part of the discrete operator.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ----------------------------------------------------------------------------
# 1-D bases (own copies: the oracle does not import the product package)
# ----------------------------------------------------------------------------

def _legendre(n, x):
    p0 = np.ones_like(x)
    if n == 0:
        return p0, np.zeros_like(x)
    p1 = x.copy()
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    with np.errstate(divide="ignore", invalid="ignore"):
        dp = n * (x * p1 - p0) / (x * x - 1.0)
    return p1, dp


def zwgl(n):
    """Gauss-Legendre nodes/weights  [UPSTREAM speclib zwgl]."""
    x, w = np.polynomial.legendre.leggauss(n)
    return x, w


def zwgll(n):
    """Gauss-Lobatto-Legendre nodes/weights  [UPSTREAM speclib zwgll]."""
    N = n - 1
    c = np.zeros(n)
    c[N] = 1.0
    xi = np.sort(np.polynomial.legendre.Legendre(c).deriv().roots().real)
    # polish with Newton on P_N'
    for _ in range(5):
        p, dp = _legendre(N, xi)
        d2p = (2 * xi * dp - N * (N + 1) * p) / (1.0 - xi * xi)
        xi = xi - dp / d2p
    x = np.concatenate([[-1.0], xi, [1.0]])
    p, _ = _legendre(N, x)
    w = 2.0 / (N * (N + 1) * p * p)
    return x, w


def interp_mat(xf, xt):
    J = np.zeros((len(xt), len(xf)))
    for j in range(len(xf)):
        num = np.ones_like(xt)
        den = 1.0
        for k in range(len(xf)):
            if k != j:
                num = num * (xt - xf[k])
                den = den * (xf[j] - xf[k])
        J[:, j] = num / den
    return J


def deriv_mat(x):
    n = len(x)
    D = np.zeros((n, n))
    c = np.array([np.prod([x[i] - x[k] for k in range(n) if k != i]) for i in range(n)])
    for i in range(n):
        for j in range(n):
            if i != j:
                D[i, j] = c[i] / (c[j] * (x[i] - x[j]))
        D[i, i] = -D[i].sum()
    return D


# BDF / EXT coefficients for constant dt, order k = min(istep,3)   [UPSTREAM setordbd/setbd/setabbd]
BD = {1: [1.0, 1.0, 0.0, 0.0], 2: [1.5, 2.0, -0.5, 0.0], 3: [11.0 / 6.0, 3.0, -1.5, 1.0 / 3.0]}
AB = {1: [1.0, 0.0, 0.0], 2: [2.0, -1.0, 0.0], 3: [3.0, -3.0, 1.0]}


class LinNS2D:
    """Linearised (direct/adjoint) incompressible Navier-Stokes time stepper,
    PnPn-2 spectral elements, BDFk/EXTk with per-matvec order ramp, 2-D."""

    def __init__(self, *, x, y, gid, nglob, mask, ub, spng, re, endtime, cfl=0.5,
                 lxd=None, has_outflow=True, build_solvers=True, factorize_pressure=True):
        self.nel, self.n = x.shape[0], x.shape[-1]
        n = self.n
        self.m = n - 2
        self.lxd = lxd if lxd else 3 * n // 2
        self.x, self.y, self.gid, self.nglob = x, y, gid.astype(np.int64), int(nglob)
        self.mask, self.ub, self.spng = mask, ub, spng
        self.nu = 1.0 / re
        self.endtime, self.cfltarget = endtime, cfl
        self.has_outflow = has_outflow

        self.z1, self.w1 = zwgll(n)
        self.z2, self.w2 = zwgl(self.m)
        self.zd, self.wd = zwgl(self.lxd)
        self.D = deriv_mat(self.z1)
        self.J12 = interp_mat(self.z1, self.z2)
        self.D12 = self.J12 @ self.D
        self.Jd = interp_mat(self.z1, self.zd)
        self.Dd = deriv_mat(self.zd)

        # geometry  [UPSTREAM coef.f geom1/geom2/glmapm1]
        dr = lambda f: f @ self.D.T          # d/dr : contract i (last index)
        ds = lambda f: self.D @ f            # d/ds : contract j
        xr, xs, yr, ys = dr(x), ds(x), dr(y), ds(y)
        self.jac = xr * ys - xs * yr
        assert self.jac.min() > 0, "non-positive Jacobian"
        self.rx, self.ry, self.sx, self.sy = ys, -xs, -yr, xr
        W = self.w1[:, None] * self.w1[None, :]
        self.bm1 = self.jac * W
        sc = W / self.jac
        self.g1 = (self.rx ** 2 + self.ry ** 2) * sc
        self.g2 = (self.sx ** 2 + self.sy ** 2) * sc
        self.g4 = (self.rx * self.sx + self.ry * self.sy) * sc
        i12 = lambda f: self.J12 @ f @ self.J12.T
        self.W2 = self.w2[:, None] * self.w2[None, :]
        self.rxm2, self.rym2, self.sxm2, self.sym2 = (i12(a) for a in (self.rx, self.ry, self.sx, self.sy))
        idl = lambda f: self.Jd @ f @ self.Jd.T
        Wd = self.wd[:, None] * self.wd[None, :]
        # set_dealias_rx: Jacobian-scaled metrics on the fine mesh times fine weights
        self.rxd, self.ryd, self.sxd, self.syd = (idl(a) * Wd for a in (self.rx, self.ry, self.sx, self.sy))

        # gather-scatter
        self.gflat = self.gid.ravel()
        self.mult = self.dssum(np.ones_like(x))
        self.binvm1 = 1.0 / self.dssum(self.bm1)
        self.volvm1 = self.bm1.sum()
        self.gmask = np.ones(self.nglob)
        np.minimum.at(self.gmask, self.gflat, mask.ravel())

        self.dt, self.nsteps = self.timestep_rule()
        self.npr = self.nel * self.m * self.m
        self._helm = {}
        self._E = None
        self._factorize_pressure = factorize_pressure     # False: assemble E only (oracle/cpu_port.py needs the matrix, not its LU)
        if build_solvers:
            self._build_pressure_solver()

    # ---------------- gather-scatter ----------------
    def dssum(self, f):
        """Direct-stiffness sum over shared nodes  [UPSTREAM dssum / gslib gs_op add]."""
        g = np.bincount(self.gflat, weights=f.ravel(), minlength=self.nglob)
        return g[self.gid]

    # ---------------- dt rule: core/matvec.f:26-46 + [UPSTREAM compute_cfl] ----------------
    def compute_cfl(self, u, v, dt):
        z = self.z1
        d = np.empty_like(z)
        d[0], d[-1] = z[1] - z[0], z[-1] - z[-2]
        d[1:-1] = 0.5 * (z[2:] - z[:-2])
        dri = 1.0 / d
        ur = (u * self.rx + v * self.ry) / self.jac
        us = (u * self.sx + v * self.sy) / self.jac
        c = np.abs(dt * ur * dri[None, None, :]) + np.abs(dt * us * dri[None, :, None])
        return c.max()

    def timestep_rule(self):
        ctarg = self.compute_cfl(self.ub[0], self.ub[1], 1.0)
        dt = self.cfltarget / ctarg
        nsteps = int(np.ceil(self.endtime / dt))
        return self.endtime / nsteps, nsteps

    # ---------------- element-local operators ----------------
    def axhelm(self, u, h1, h2):
        """w = h1 * D^T G D u + h2 * B u (local, unassembled)  [UPSTREAM hmholtz.f axhelm]."""
        ur, us = u @ self.D.T, self.D @ u
        t1 = self.g1 * ur + self.g4 * us
        t2 = self.g2 * us + self.g4 * ur
        return h1 * (t1 @ self.D + self.D.T @ t2) + h2 * self.bm1 * u

    def opdiv(self, u, v):
        """D u : weak divergence GLL -> GL  [UPSTREAM navier1.f opdiv/multd]."""
        def dcomp(f, rm, sm):
            fr = self.J12 @ f @ self.D12.T       # d/dr on mesh 2
            fs = self.D12 @ f @ self.J12.T
            return (rm * fr + sm * fs) * self.W2
        return dcomp(u, self.rxm2, self.sxm2) + dcomp(v, self.rym2, self.sym2)

    def opgradt(self, p):
        """D^T p (local)  [UPSTREAM navier1.f opgradt/cdtp]."""
        wp = p * self.W2
        def tcomp(rm, sm):
            return self.J12.T @ (wp * rm) @ self.D12 + self.D12.T @ (wp * sm) @ self.J12
        return tcomp(self.rxm2, self.sxm2), tcomp(self.rym2, self.sym2)

    def convect(self, cx, cy, phi):
        """J^T [ w_d (c . grad phi) ] on the lxd Gauss mesh  [UPSTREAM convect.f convect_new];
        result is the *mass-weighted* term (convop divides by bm1 and advabp multiplies back)."""
        fx, fy = self.Jd @ cx @ self.Jd.T, self.Jd @ cy @ self.Jd.T
        pf = self.Jd @ phi @ self.Jd.T
        pr, ps = pf @ self.Dd.T, self.Dd @ pf
        cr = self.rxd * fx + self.ryd * fy
        cs = self.sxd * fx + self.syd * fy
        return self.Jd.T @ (cr * pr + cs * ps) @ self.Jd

    def convect_adj(self, cx, cy, Ux, Uy):
        """J^T [ w_d (grad U)^T c ]  [UPSTREAM convect.f convect_adj]."""
        fx, fy = self.Jd @ cx @ self.Jd.T, self.Jd @ cy @ self.Jd.T
        ox = np.zeros_like(fx)
        oy = np.zeros_like(fx)
        for f, U in ((fx, Ux), (fy, Uy)):
            uf = self.Jd @ U @ self.Jd.T
            ur, us = uf @ self.Dd.T, self.Dd @ uf
            ox += f * (self.rxd * ur + self.sxd * us)
            oy += f * (self.ryd * ur + self.syd * us)
        return self.Jd.T @ ox @ self.Jd, self.Jd.T @ oy @ self.Jd

    # ---------------- sparse direct solvers (stand-in for PCG / GMRES+Schwarz) ----------------
    def _local_matrices(self, op, nin):
        """Dense element matrices of a linear element-local operator by probing."""
        eye = np.eye(nin * nin).reshape(nin * nin, 1, nin, nin)
        out = op(np.broadcast_to(eye, (nin * nin, self.nel, nin, nin)))
        nout = out.shape[-1]
        # -> (nel, nout^2, nin^2)
        return out.reshape(nin * nin, self.nel, nout * nout).transpose(1, 2, 0)

    def _helm_solver(self, h1, h2):
        key = (h1, h2)
        if key not in self._helm:
            n = self.n
            K = self._local_matrices(lambda u: self.axhelm(u, h1, h2), n)
            g = self.gid.reshape(self.nel, n * n)
            rows = np.repeat(g[:, :, None], n * n, axis=2).ravel()
            cols = np.repeat(g[:, None, :], n * n, axis=1).ravel()
            A = sp.coo_matrix((K.ravel(), (rows, cols)), shape=(self.nglob, self.nglob)).tocsr()
            free = np.where(self.gmask > 0)[0]
            self._free = free
            self._helm[key] = spla.splu(A[free][:, free].tocsc())
        return self._helm[key]

    def helm_solve(self, r_local, h1, h2):
        """Solve H du = dssum(r) with homogeneous Dirichlet mask; returns local copy."""
        lu = self._helm_solver(h1, h2)
        rg = np.bincount(self.gflat, weights=r_local.ravel(), minlength=self.nglob)
        ug = np.zeros(self.nglob)
        ug[self._free] = lu.solve(rg[self._free])
        return ug[self.gid]

    def _build_pressure_solver(self):
        n, m = self.n, self.m
        g = self.gid.reshape(self.nel, n * n)
        prow = np.arange(self.npr).reshape(self.nel, m * m)
        rows = np.repeat(prow[:, :, None], n * n, axis=2).ravel()
        cols = np.repeat(g[:, None, :], m * m, axis=1).ravel()
        zero = lambda u: np.zeros_like(u)
        Gs = []
        for c in range(2):
            op = (lambda u: self.opdiv(u, zero(u))) if c == 0 else (lambda u: self.opdiv(zero(u), u))
            Dl = self._local_matrices(op, n)       # (nel, m^2, n^2)
            Gs.append(sp.coo_matrix((Dl.ravel(), (rows, cols)), shape=(self.npr, self.nglob)).tocsr())
        binv_g = np.zeros(self.nglob)
        binv_g[self.gflat] = self.binvm1.ravel()
        Wg = sp.diags(binv_g * self.gmask)
        E = sum(G @ Wg @ G.T for G in Gs).tocsc()       # D B^-1 D^T (without 1/h2)
        self._Emat = E
        if not self._factorize_pressure:
            return
        if self.has_outflow:
            self._E = spla.splu(E)
        else:
            # constant null space: pin dof 0 (consistent rhs after `ortho`), then re-centre
            self._E = spla.splu(E[1:, 1:].tocsc())

    def E_solve(self, g):
        """Solve (D B^-1 D^T) x = g   (zero-mean solution when E is singular)."""
        if self.has_outflow:
            return self._E.solve(g.ravel()).reshape(g.shape)
        gg = g.ravel() - g.mean()                      # [UPSTREAM navier1.f ortho]
        x = np.concatenate([[0.0], self._E.solve(gg[1:])])
        return (x - x.mean()).reshape(g.shape)

    # ---------------- one nek_advance() in perturbation mode ----------------
    def new_state(self, q):
        u, v, p = q
        z = lambda a: np.zeros_like(a)
        return dict(u=u.copy(), v=v.copy(), p=p.copy(),
                    ulag=[[z(u), z(v)], [z(u), z(v)]],
                    exlag=[[z(u), z(v)], [z(u), z(v)]], plag=z(p))

    def step(self, st, istep, adjoint=False):
        k = min(istep, 3)
        bd, ab = BD[k], AB[k]
        dt = self.dt
        u, v, p = st["u"], st["v"], st["p"]
        U, V = self.ub
        # makeufp: user forcing (sponge), mass weighted
        bfx = -self.spng * u * self.bm1
        bfy = -self.spng * v * self.bm1
        # advabp / advabp_adjoint
        if not adjoint:
            bfx -= self.convect(u, v, U) + self.convect(U, V, u)
            bfy -= self.convect(u, v, V) + self.convect(U, V, v)
        else:
            ax, ay = self.convect_adj(u, v, U, V)
            bfx += -ax + self.convect(U, V, u)
            bfy += -ay + self.convect(U, V, v)
        # makextp
        ex = st["exlag"]
        tx = ab[1] * ex[0][0] + ab[2] * ex[1][0]
        ty = ab[1] * ex[0][1] + ab[2] * ex[1][1]
        ex[1] = ex[0]
        ex[0] = [bfx, bfy]
        bfx = ab[0] * bfx + tx
        bfy = ab[0] * bfy + ty
        # makebdfp
        ul = st["ulag"]
        bfx = bfx + self.bm1 * (bd[1] * u + bd[2] * ul[0][0] + bd[3] * ul[1][0]) / dt
        bfy = bfy + self.bm1 * (bd[1] * v + bd[2] * ul[0][1] + bd[3] * ul[1][1]) / dt
        # lagfieldp
        ul[1] = ul[0]
        ul[0] = [u, v]
        # perturbv(igeom=2)
        h1, h2 = self.nu, bd[0] / dt
        pext = p if k < 3 else 2.0 * p - st["plag"]         # extrapprp
        gx, gy = self.opgradt(pext)
        rxx = bfx + gx - self.axhelm(u, h1, h2)              # cresvipp
        ryy = bfy + gy - self.axhelm(v, h1, h2)
        us = u + self.helm_solve(rxx, h1, h2)                # ophinv
        vs = v + self.helm_solve(ryy, h1, h2)
        # incomprp:  E dp = -D u*,  E = D (h2 B)^-1 D^T
        dp = self.E_solve(-self.opdiv(us, vs)) * h2
        wx, wy = self.opgradt(dp)
        fac = self.binvm1 * self.mask / h2
        st["u"] = us + fac * self.dssum(wx * self.mask)
        st["v"] = vs + fac * self.dssum(wy * self.mask)
        st["plag"] = p                                       # lagpresp
        st["p"] = pext + dp
        return st

    # ---------------- nonlinear Navier-Stokes step (newton_krylov's nonlinear_forward_map) ----------------
    def set_baseflow(self, ub):
        """New linearisation point (core/newton_krylov.f:371-372) and the dt / nsteps that
        prepare_linearized_solver derives from it (core/matvec.f:26-46)."""
        self.ub = np.array(ub)
        self.dt, self.nsteps = self.timestep_rule()
        self._helm = {}

    def step_nonlinear(self, st, istep, spng_str=0.0, spng_vr=None):
        """One nek_advance() of the full equations: makef (user forcing, -B (u.grad)u dealiased, EXT, BDF)
        then the same Helmholtz / pressure projection as the perturbation step  [UPSTREAM plan3, makef,
        advab, makeabf, makebdf, cresvif, incomprn].  Dirichlet values ride along in u (masked solves)."""
        k = min(istep, 3)
        bd, ab = BD[k], AB[k]
        dt = self.dt
        u, v, p = st["u"], st["v"], st["p"]
        bfx = -self.convect(u, v, u)
        bfy = -self.convect(u, v, v)
        if spng_str != 0.0:                                   # nekStab_forcing, DNS branch (core/utils.f:165-170)
            bfx += self.spng * (spng_vr[0] - u) * spng_str * self.bm1
            bfy += self.spng * (spng_vr[1] - v) * spng_str * self.bm1
        ex = st["exlag"]
        tx = ab[1] * ex[0][0] + ab[2] * ex[1][0]
        ty = ab[1] * ex[0][1] + ab[2] * ex[1][1]
        ex[1] = ex[0]
        ex[0] = [bfx, bfy]
        bfx = ab[0] * bfx + tx
        bfy = ab[0] * bfy + ty
        ul = st["ulag"]
        bfx = bfx + self.bm1 * (bd[1] * u + bd[2] * ul[0][0] + bd[3] * ul[1][0]) / dt
        bfy = bfy + self.bm1 * (bd[1] * v + bd[2] * ul[0][1] + bd[3] * ul[1][1]) / dt
        ul[1] = ul[0]
        ul[0] = [u, v]
        h1, h2 = self.nu, bd[0] / dt
        pext = p if k < 3 else 2.0 * p - st["plag"]
        gx, gy = self.opgradt(pext)
        us = u + self.helm_solve(bfx + gx - self.axhelm(u, h1, h2), h1, h2)
        vs = v + self.helm_solve(bfy + gy - self.axhelm(v, h1, h2), h1, h2)
        dp = self.E_solve(-self.opdiv(us, vs)) * h2
        wx, wy = self.opgradt(dp)
        fac = self.binvm1 * self.mask / h2
        st["u"] = us + fac * self.dssum(wx * self.mask)
        st["v"] = vs + fac * self.dssum(wy * self.mask)
        st["plag"] = p
        st["p"] = pext + dp
        return st

    def nonlinear_map(self, q, nsteps=None, spng_str=0.0):
        """Phi_T(q): nsteps of the full equations from q (core/newton_krylov.f:336-378)."""
        st = self.new_state(q)
        vr = (q[0].copy(), q[1].copy())
        for istep in range(1, (nsteps or self.nsteps) + 1):
            st = self.step_nonlinear(st, istep, spng_str, vr)
        return st["u"], st["v"], st["p"]

    def matvec(self, q, adjoint=False, nsteps=None):
        """f = Phi_T q  (core/matvec.f:163-243 / :249-326)."""
        st = self.new_state(q)
        for istep in range(1, (nsteps or self.nsteps) + 1):
            st = self.step(st, istep, adjoint)
        return st["u"], st["v"], st["p"]

    # ---------------- Krylov vector algebra (core/krylov_subspace.f) ----------------
    def bm1s(self):
        b = self.bm1.copy()
        b[self.spng != 0] = 0.0                              # core/usr_extra.f:116-118
        return b

    def inner(self, a, b, w=None):
        w = self.bm1s() if w is None else w
        return float(np.sum(a[0] * w * b[0]) + np.sum(a[1] * w * b[1]))


def arnoldi(op: LinNS2D, q0, k, adjoint=False, log=None):
    """k-step Arnoldi with the reference's two-pass MGS
    (core/krylov_decomposition.f:7-104, :116-202). Returns (Q list, H (k+1,k))."""
    w = op.bm1s()
    nrm = np.sqrt(op.inner(q0, q0, w))
    Q = [tuple(c / nrm for c in q0)]
    H = np.zeros((k + 1, k))
    for j in range(k):
        f = list(op.matvec(Q[j], adjoint))
        for _pass in range(2):
            for i in range(j + 1):
                a = op.inner(f, Q[i], w)
                f = [fc - a * qc for fc, qc in zip(f, Q[i])]
                H[i, j] += a
        beta = np.sqrt(op.inner(f, f, w))
        H[j + 1, j] = beta
        Q.append(tuple(c / beta for c in f))
        if log:
            log(j, H)
    return Q, H


def ritz(H, k=None):
    """eig of H(1:k,1:k) sorted by decreasing modulus + residuals
    |H(k+1,k) y_k|  (core/eigensolvers.f:346-350, core/lapack_wrapper.f:129-251)."""
    k = k or H.shape[1]
    vals, vecs = np.linalg.eig(H[:k, :k])
    order = np.argsort(-np.abs(vals), kind="stable")
    vals, vecs = vals[order], vecs[:, order]
    res = np.abs(H[k, k - 1] * vecs[k - 1, :])
    return vals, vecs, res
