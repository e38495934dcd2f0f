"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (3-D restatement).

Same role and rules as ``oracle/linns.py`` (only tests/, smoke() and bench's cpu_baseline may
import it).  Restates the perturbation-mode ``nek_advance`` of Nek5000 for hexahedral elements:
the reference drives it identically in 2-D and 3-D (core/matvec.f:163-326 loops ``nek_advance``;
the ``krylov_vector`` carries ``vz``, core/krylov_subspace.f:7-15; the inner product adds the
third component ``if3d``, core/krylov_subspace.f:41-43).  [UPSTREAM] 3-D branches of geom1/geom2
(coef.f), axhelm (hmholtz.f), multd/cdtp (navier1.f), convect_new / convect_adj (convect.f).

Parity pin: on a z-extruded mesh a z-invariant state must step exactly like the pinned 2-D
oracle (tests/test_oracle3d.py); there are no 3-D golden files in the reference.
Arrays are ``[e, k, j, i]``; solves are sparse-direct.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .linns import AB, BD, deriv_mat, interp_mat, zwgl, zwgll


def ax(A, f, axis):
    """Apply the matrix A (n_out x n_in) along ``axis`` of f (-1: r, -2: s, -3: t)."""
    return np.moveaxis(np.tensordot(A, f, axes=(1, axis)), 0, axis)


class LinNS3D:
    def __init__(self, *, x, y, z, gid, nglob, mask, ub, spng, re, endtime, cfl=0.5, lxd=None,
                 has_outflow=True, build_solvers=True):
        self.nel, self.n = x.shape[0], x.shape[-1]
        n = self.n
        self.m = n - 2
        self.lxd = lxd if lxd else 3 * n // 2
        self.x, self.y, self.z = x, y, z
        self.gid, self.nglob = gid.astype(np.int64), int(nglob)
        self.mask, self.ub, self.spng = mask, np.array(ub), spng
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
        D = self.D
        # geometry  [UPSTREAM coef.f geom1 (3-D branch): Jacobian-scaled inverse metrics]
        g = {}
        for nm, f in (("x", x), ("y", y), ("z", z)):
            g[nm + "r"], g[nm + "s"], g[nm + "t"] = ax(D, f, -1), ax(D, f, -2), ax(D, f, -3)
        xr, xs, xt, yr, ys, yt, zr, zs, zt = (g[k] for k in ("xr", "xs", "xt", "yr", "ys", "yt", "zr", "zs", "zt"))
        self.jac = xr * ys * zt + xt * yr * zs + xs * yt * zr - xr * yt * zs - xs * yr * zt - xt * ys * zr
        assert self.jac.min() > 0, "non-positive Jacobian"
        # met[a][c] = J * d(xi_a)/d(x_c),  a in (r,s,t), c in (x,y,z)
        self.met = [[ys * zt - yt * zs, xt * zs - xs * zt, xs * yt - xt * ys],
                    [yt * zr - yr * zt, xr * zt - xt * zr, xt * yr - xr * yt],
                    [yr * zs - ys * zr, xs * zr - xr * zs, xr * ys - xs * yr]]
        W = self.w1[:, None, None] * self.w1[None, :, None] * self.w1[None, None, :]
        self.bm1 = self.jac * W
        sc = W / self.jac
        dotm = lambda a, b: sum(self.met[a][c] * self.met[b][c] for c in range(3)) * sc
        # [UPSTREAM geom2]: G1..G6 = rr, ss, tt, rs, rt, st
        self.G = [[dotm(0, 0), dotm(0, 1), dotm(0, 2)], [dotm(0, 1), dotm(1, 1), dotm(1, 2)], [dotm(0, 2), dotm(1, 2), dotm(2, 2)]]
        i12 = lambda f: ax(self.J12, ax(self.J12, ax(self.J12, f, -1), -2), -3)
        self.W2 = self.w2[:, None, None] * self.w2[None, :, None] * self.w2[None, None, :]
        self.met2 = [[i12(self.met[a][c]) for c in range(3)] for a in range(3)]
        idl = lambda f: ax(self.Jd, ax(self.Jd, ax(self.Jd, f, -1), -2), -3)
        Wd = self.wd[:, None, None] * self.wd[None, :, None] * self.wd[None, None, :]
        self.metd = [[idl(self.met[a][c]) * Wd for c in range(3)] for a in range(3)]      # set_dealias_rx
        self.gflat = self.gid.ravel()
        self.mult = self.dssum(np.ones_like(x))
        self.binvm1 = 1.0 / self.dssum(self.bm1)
        self.volvm1 = self.bm1.sum()
        self.gmask = np.ones(self.nglob)
        np.minimum.at(self.gmask, self.gflat, mask.ravel())
        self.dt, self.nsteps = self.timestep_rule()
        self.npr = self.nel * self.m ** 3
        self._helm = {}
        self._E = None
        if build_solvers:
            self._build_pressure_solver()

    def dssum(self, f):
        return np.bincount(self.gflat, weights=f.ravel(), minlength=self.nglob)[self.gid]

    # ---------------- dt rule (core/matvec.f:26-46, [UPSTREAM compute_cfl] 3-D branch) ----------------
    def compute_cfl(self, u, dt):
        zz = self.z1
        d = np.empty_like(zz)
        d[0], d[-1] = zz[1] - zz[0], zz[-1] - zz[-2]
        d[1:-1] = 0.5 * (zz[2:] - zz[:-2])
        dri = 1.0 / d
        c = 0.0
        shp = [(1, 1, 1, -1), (1, 1, -1, 1), (1, -1, 1, 1)]
        for a in range(3):
            ua = sum(u[cc] * self.met[a][cc] for cc in range(3)) / self.jac
            c = c + np.abs(dt * ua * dri.reshape(shp[a]))
        return c.max()

    def timestep_rule(self):
        ctarg = self.compute_cfl(self.ub, 1.0)
        dt = self.cfltarget / ctarg
        nsteps = int(np.ceil(self.endtime / dt))
        return self.endtime / nsteps, nsteps

    # ---------------- element-local operators ----------------
    def grad_rst(self, u, D):
        return [ax(D, u, -1), ax(D, u, -2), ax(D, u, -3)]

    def axhelm(self, u, h1, h2):
        ur = self.grad_rst(u, self.D)
        t = [sum(self.G[a][b] * ur[b] for b in range(3)) for a in range(3)]
        DT = self.D.T
        return h1 * (ax(DT, t[0], -1) + ax(DT, t[1], -2) + ax(DT, t[2], -3)) + h2 * self.bm1 * u

    def _to2(self, f, which):
        """d/d(xi_which) of f evaluated on mesh 2 (which = 0,1,2 for r,s,t)."""
        mats = [self.D12 if which == a else self.J12 for a in range(3)]
        return ax(mats[2], ax(mats[1], ax(mats[0], f, -1), -2), -3)

    def opdiv(self, u):
        out = 0.0
        for c in range(3):
            for a in range(3):
                out = out + self.met2[a][c] * self._to2(u[c], a)
        return out * self.W2

    def opgradt(self, p):
        wp = p * self.W2
        out = []
        for c in range(3):
            g = 0.0
            for a in range(3):
                mats = [(self.D12 if a == b else self.J12).T for b in range(3)]
                g = g + ax(mats[2], ax(mats[1], ax(mats[0], wp * self.met2[a][c], -1), -2), -3)
            out.append(g)
        return out

    def _fine(self, f):
        return ax(self.Jd, ax(self.Jd, ax(self.Jd, f, -1), -2), -3)

    def _coarse(self, f):
        JT = self.Jd.T
        return ax(JT, ax(JT, ax(JT, f, -1), -2), -3)

    def convect(self, c, phi):
        """J^T [ w_d (c . grad phi) ]  [UPSTREAM convect_new]."""
        cf = [self._fine(cc) for cc in c]
        pr = self.grad_rst(self._fine(phi), self.Dd)
        out = 0.0
        for a in range(3):
            ca = sum(self.metd[a][cc] * cf[cc] for cc in range(3))
            out = out + ca * pr[a]
        return self._coarse(out)

    def convect_adj(self, c, U):
        """J^T [ w_d (grad U)^T c ]  [UPSTREAM convect_adj]."""
        cf = [self._fine(cc) for cc in c]
        o = [0.0, 0.0, 0.0]
        for cc in range(3):
            ur = self.grad_rst(self._fine(U[cc]), self.Dd)
            for xx in range(3):
                o[xx] = o[xx] + cf[cc] * sum(self.metd[a][xx] * ur[a] for a in range(3))
        return [self._coarse(q) for q in o]

    # ---------------- sparse direct solvers ----------------
    def _local_matrices(self, op, nin, chunk=64):
        nn = nin ** 3
        outs = []
        for k0 in range(0, nn, chunk):
            k1 = min(nn, k0 + chunk)
            eye = np.zeros((k1 - k0, 1, nn))
            eye[np.arange(k1 - k0), 0, np.arange(k0, k1)] = 1.0
            inp = np.broadcast_to(eye.reshape(k1 - k0, 1, nin, nin, nin), (k1 - k0, self.nel, nin, nin, nin))
            o = op(inp)
            outs.append(o.reshape(k1 - k0, self.nel, -1))
        out = np.concatenate(outs, axis=0)                 # (nn_in, nel, nn_out)
        return out.transpose(1, 2, 0)                      # (nel, nn_out, nn_in)

    def _helm_solver(self, h1, h2):
        key = (h1, h2)
        if key not in self._helm:
            nn = self.n ** 3
            K = self._local_matrices(lambda u: self.axhelm(u, h1, h2), self.n)
            g = self.gid.reshape(self.nel, nn)
            rows = np.repeat(g[:, :, None], nn, axis=2).ravel()
            cols = np.repeat(g[:, None, :], nn, axis=1).ravel()
            A = sp.coo_matrix((K.ravel(), (rows, cols)), shape=(self.nglob, self.nglob)).tocsr()
            free = np.where(self.gmask > 0)[0]
            self._free = free
            self._helm[key] = spla.splu(A[free][:, free].tocsc())
        return self._helm[key]

    def helm_solve(self, r_local, h1, h2):
        lu = self._helm_solver(h1, h2)
        rg = np.bincount(self.gflat, weights=r_local.ravel(), minlength=self.nglob)
        ug = np.zeros(self.nglob)
        ug[self._free] = lu.solve(rg[self._free])
        return ug[self.gid]

    def _build_pressure_solver(self):
        n, m = self.n, self.m
        nn, mm = n ** 3, m ** 3
        g = self.gid.reshape(self.nel, nn)
        prow = np.arange(self.npr).reshape(self.nel, mm)
        rows = np.repeat(prow[:, :, None], nn, axis=2).ravel()
        cols = np.repeat(g[:, None, :], mm, axis=1).ravel()
        Gs = []
        for c in range(3):
            def op(u, c=c):
                z = np.zeros_like(u)
                v = [z, z, z]
                v[c] = u
                return self.opdiv(v)
            Dl = self._local_matrices(op, n)
            Gs.append(sp.coo_matrix((Dl.ravel(), (rows, cols)), shape=(self.npr, self.nglob)).tocsr())
        binv_g = np.zeros(self.nglob)
        binv_g[self.gflat] = self.binvm1.ravel()
        Wg = sp.diags(binv_g * self.gmask)
        E = sum(G @ Wg @ G.T for G in Gs).tocsc()
        self._Emat = E
        self._E = spla.splu(E) if self.has_outflow else spla.splu(E[1:, 1:].tocsc())

    def E_solve(self, g):
        if self.has_outflow:
            return self._E.solve(g.ravel()).reshape(g.shape)
        gg = g.ravel() - g.mean()
        xx = np.concatenate([[0.0], self._E.solve(gg[1:])])
        return (xx - xx.mean()).reshape(g.shape)

    # ---------------- one nek_advance() in perturbation mode ----------------
    def new_state(self, q):
        u, p = [q[0], q[1], q[2]], q[3]
        zz = lambda: [np.zeros_like(u[0]) for _ in range(3)]
        return dict(u=[a.copy() for a in u], p=p.copy(), ulag=[zz(), zz()], exlag=[zz(), zz()], plag=np.zeros_like(p))

    def _finish(self, st, bf, k, bd):
        dt = self.dt
        u, p = st["u"], st["p"]
        h1, h2 = self.nu, bd[0] / dt
        pext = p if k < 3 else 2.0 * p - st["plag"]
        gp = self.opgradt(pext)
        us = [u[c] + self.helm_solve(bf[c] + gp[c] - self.axhelm(u[c], h1, h2), h1, h2) for c in range(3)]
        dp = self.E_solve(-self.opdiv(us)) * h2
        w = self.opgradt(dp)
        fac = self.binvm1 * self.mask / h2
        st["u"] = [us[c] + fac * self.dssum(w[c] * self.mask) for c in range(3)]
        st["plag"] = p
        st["p"] = pext + dp
        return st

    def _ext_bdf(self, st, bf, k):
        bd, ab = BD[k], AB[k]
        u = st["u"]
        ex, ul = st["exlag"], st["ulag"]
        out = []
        for c in range(3):
            out.append(ab[0] * bf[c] + ab[1] * ex[0][c] + ab[2] * ex[1][c]
                       + self.bm1 * (bd[1] * u[c] + bd[2] * ul[0][c] + bd[3] * ul[1][c]) / self.dt)
        st["exlag"] = [list(bf), ex[0]]
        st["ulag"] = [list(u), ul[0]]
        return out, bd

    def step(self, st, istep, adjoint=False):
        k = min(istep, 3)
        u, U = st["u"], self.ub
        bf = [-self.spng * u[c] * self.bm1 for c in range(3)]
        if not adjoint:
            for c in range(3):
                bf[c] = bf[c] - (self.convect(u, U[c]) + self.convect(U, u[c]))
        else:
            a = self.convect_adj(u, U)
            for c in range(3):
                bf[c] = bf[c] - a[c] + self.convect(U, u[c])
        bf, bd = self._ext_bdf(st, bf, k)
        return self._finish(st, bf, k, bd)

    def step_nonlinear(self, st, istep):
        k = min(istep, 3)
        u = st["u"]
        bf = [-self.convect(u, u[c]) for c in range(3)]
        bf, bd = self._ext_bdf(st, bf, k)
        return self._finish(st, bf, k, bd)

    def matvec(self, q, adjoint=False, nsteps=None):
        st = self.new_state(q)
        for istep in range(1, (nsteps or self.nsteps) + 1):
            st = self.step(st, istep, adjoint)
        return st["u"][0], st["u"][1], st["u"][2], st["p"]

    def nonlinear_map(self, q, nsteps=None):
        st = self.new_state(q)
        for istep in range(1, (nsteps or self.nsteps) + 1):
            st = self.step_nonlinear(st, istep)
        return st["u"][0], st["u"][1], st["u"][2], st["p"]

    def bm1s(self):
        b = self.bm1.copy()
        b[self.spng != 0] = 0.0
        return b

    def inner(self, a, b, w=None):
        w = self.bm1s() if w is None else w
        return float(sum(np.sum(a[c] * w * b[c]) for c in range(3)))
