"""Arnoldi checkpoint / restart (nekstab_amd/checkpoint.py: arnoldi_checkpoint core/eigensolvers.f:802-905, the restart
branch :284-325) on the CPU: a numpy backend with the device vector interface on the real cylinder mesh; the same files
through the HIP library are exercised in tests/test_krylov_gpu.py."""
import os

import numpy as np

from nekstab_amd import checkpoint, krylov, nekio


class FieldBackend:
    """Vectors = (vx, vy, pr) on the case's meshes; the 'time stepper' is a fixed linear map on the velocity (a weighted
    neighbour average inside every element plus a rotation of the two components), the inner product is mass weighted and
    excludes the pressure -- what krylov_inner_product does."""

    def __init__(self, case, w):
        self.case, self.w = case, w
        self.nsteps = 7
        self.dt = 0.25
        n = case.lx1
        rng = np.random.default_rng(4)
        self.S = np.eye(n) * 0.6 + 0.4 * rng.random((n, n)) / n
        self.th = 0.3

    def alloc(self, k=1):
        m = self.case.lx1 - 2
        return [[np.zeros(self.case.x.shape), np.zeros(self.case.x.shape), np.zeros((self.case.nel, m, m))] for _ in range(k)]

    def upload(self, v, vx, vy, pr):
        v[0], v[1], v[2] = np.array(vx, dtype=float), np.array(vy, dtype=float), np.array(pr, dtype=float)

    def download(self, v):
        return v[0].copy(), v[1].copy(), v[2].copy()

    def matvec(self, f, q, mode=0):
        a = np.einsum("ij,ejk,lk->eil", self.S, q[0], self.S)
        b = np.einsum("ij,ejk,lk->eil", self.S, q[1], self.S)
        c, s = np.cos(self.th), np.sin(self.th)
        f[0], f[1], f[2] = c * a - s * b, s * a + c * b, 0.5 * q[2]

    def dot(self, p, q):
        return float(np.sum(self.w * (p[0] * q[0] + p[1] * q[1])))

    def norm(self, p):
        return np.sqrt(self.dot(p, p))

    def scal(self, p, a):
        for k in range(3):
            p[k] = p[k] * a

    def orth(self, f, Q):
        h = np.zeros(len(Q))
        for _ in range(2):
            c = np.array([self.dot(f, q) for q in Q])
            for ci, q in zip(c, Q):
                for k in range(3):
                    f[k] = f[k] - ci * q[k]
            h += c
        beta = self.norm(f)
        self.scal(f, 1.0 / beta)
        return h, beta


def test_checkpoint_files_restart_the_same_factorisation(tmp_path, case6, oracle6_nosolve):
    be = FieldBackend(case6, oracle6_nosolve.bm1)
    rng = np.random.default_rng(8)
    k_dim = 7
    Q = be.alloc(k_dim + 1)
    be.upload(Q[0], rng.standard_normal(case6.x.shape), rng.standard_normal(case6.x.shape), rng.standard_normal((case6.nel, 4, 4)))
    be.scal(Q[0], 1.0 / be.norm(Q[0]))
    H = np.zeros((k_dim + 1, k_dim))
    out = str(tmp_path)
    for k in range(1, k_dim + 1):
        krylov.arnoldi_factorization(be, Q, H, k, k)
        checkpoint.arnoldi_checkpoint(be, case6, Q, H, k, out, session="1cyl", evop="d", sampling_period=2.0)
    # the files the reference's restart reads
    for k in range(1, k_dim + 1):
        assert os.path.exists(os.path.join(out, "HES1cyl%04d" % k)) and os.path.exists(os.path.join(out, "Spectre_Hd%04d.dat" % k))
    assert sorted(f for f in os.listdir(out) if f.startswith("KRY")) == ["KRY1cyl0.f%05d" % i for i in range(1, k_dim + 2)]
    sp = np.loadtxt(os.path.join(out, "Spectre_Hd%04d.dat" % k_dim))
    vals, _ = krylov.eig_sorted(H[:k_dim, :k_dim])
    assert np.abs(sp[:, 0] + 1j * sp[:, 1] - vals).max() < 1e-6              # (3E15.7)
    # restart after step 3 (uparam(2) = 3) and finish: the same Hessenberg matrix and the same last vector
    Q2, H2, mstart = checkpoint.load_checkpoint(be, case6, out, k_dim, 3, session="1cyl")
    assert mstart == 4 and np.array_equal(H2[:4, :3], H[:4, :3])
    krylov.arnoldi_factorization(be, Q2, H2, mstart, k_dim)
    assert np.abs(H2 - H).max() < 1e-11 * np.abs(H).max()
    for a, b in zip(Q2[k_dim][:2], Q[k_dim][:2]):
        assert np.abs(a - b).max() < 1e-10
    # k_dim < mstart, the reference's "subsampling" branch (core/eigensolvers.f:295-301): a run with k_dim = 4 restarts from
    # the checkpoint of step 6 -> the leading 5 x 4 block and the first 5 vectors: a complete factorisation of size 4
    Q3, H3, m3 = checkpoint.load_checkpoint(be, case6, out, 4, 6, session="1cyl")
    assert m3 == 5 and H3.shape == (5, 4) and np.array_equal(H3, H[:5, :4])
    for i in range(5):
        for a, b in zip(Q3[i][:2], Q[i][:2]):
            assert np.abs(a - b).max() < 1e-12
    # ... and it IS an Arnoldi relation: M Q_4 = Q_5 H(5, 4) (column 4, through the backend's own map)
    f4 = be.alloc(1)[0]
    be.matvec(f4, Q3[3], 0)
    for k in range(2):
        assert np.abs(f4[k] - sum(H3[i, 3] * Q3[i][k] for i in range(5))).max() < 1e-9 * max(1.0, np.abs(f4[k]).max())
    # the pressure travels on mesh 1 in the file and comes back on mesh 2 (map21 then map12: exact for its polynomial degree)
    v = be.alloc(1)[0]
    checkpoint.read_krylov_vector(be, case6, v, os.path.join(out, "KRY1cyl0.f00001"))
    assert np.abs(v[2] - Q[0][2]).max() < 1e-12 * max(1.0, np.abs(Q[0][2]).max())
    f = nekio.read_fld(os.path.join(out, "KRY1cyl0.f00003"))
    assert abs(f.time - 2.0) < 1e-12 and f.u.shape[0] == 2


class FieldBackend2(FieldBackend):
    def copy(self, dst, src):
        for k in range(3):
            dst[k] = src[k].copy()

    def free(self, vs):
        pass

    def basis_gemv(self, Q, y, re, im=None):
        for k in range(3):
            M = np.stack([q[k] for q in Q], axis=-1)
            re[k] = M @ np.real(y)
            if im is not None:
                im[k] = M @ np.imag(y)


def test_krylov_schur_and_outpost_files(tmp_path, case6, oracle6_nosolve):
    """krylov_schur -> outpost_ks (core/eigensolvers.f:141-388, :502-727) end to end on the numpy backend: spectra tables,
    eigenmode field files of the converged pairs with the reference's names and normalisation, the .info manifest."""
    from nekstab_amd import outpost
    be = FieldBackend2(case6, oracle6_nosolve.bm1)
    rng = np.random.default_rng(9)
    q0 = be.alloc(1)[0]
    be.upload(q0, rng.standard_normal(case6.x.shape), rng.standard_normal(case6.x.shape), np.zeros((case6.nel, 4, 4)))
    res = krylov.krylov_schur(be, q0, 40, mode=0, schur_tgt=0, eigen_tol=1e-8)
    nconv = int(np.sum(res.residual < 1e-8))
    assert nconv >= 2 and res.matvecs == 40
    out = str(tmp_path)
    files = outpost.outpost_ks(be, res, case6, out, evop="d", sampling_period=2.0, eigen_tol=1e-8, maxmodes=4, session="1cyl", wdsize=8)
    nmodes = min(nconv, 4)
    assert [os.path.basename(f) for f in files] == [n for i in range(1, nmodes + 1) for n in ("dRe1cyl0.f%05d" % i, "dIm1cyl0.f%05d" % i)]
    sp = np.loadtxt(os.path.join(out, "Spectre_Hd.dat"))
    assert sp.shape == (40, 3) and np.abs(sp[:, 0] + 1j * sp[:, 1] - res.vals).max() < 1e-6
    ns = np.loadtxt(os.path.join(out, "Spectre_NSd.dat"))
    assert np.abs(ns[:, 0] + 1j * ns[:, 1] - np.log(res.vals.astype(complex)) / 2.0).max() < 1e-6
    info = outpost.read_info(os.path.join(out, "Spectre_d.info"))
    assert info["nsteps"] == "7" and info["k_dim"] == "40" and int(info["outposted"]) == nmodes
    # the first outposted mode is an eigenvector of the backend's map: M (re + i im) = mu (re + i im), unit norm
    i0 = int(np.argmax(res.residual < 1e-8))
    re, im = be.alloc(2)
    fr, fi = nekio.read_fld(files[0]), nekio.read_fld(files[1])
    be.upload(re, fr.u[0, :, 0], fr.u[1, :, 0], np.zeros((case6.nel, 4, 4)))
    be.upload(im, fi.u[0, :, 0], fi.u[1, :, 0], np.zeros((case6.nel, 4, 4)))
    assert abs(be.dot(re, re) + be.dot(im, im) - 1.0) < 1e-10
    mre, mim = be.alloc(2)
    be.matvec(mre, re); be.matvec(mim, im)
    mu = res.vals[i0]
    err = 0.0
    for c in range(2):
        lhs = mre[c] + 1j * mim[c]
        rhs = mu * (re[c] + 1j * im[c])
        err = max(err, np.abs(lhs - rhs).max())
    assert err < 1e-6
