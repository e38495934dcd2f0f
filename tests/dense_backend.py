"""A tiny dense numpy 'device' with the NekStabHip vector interface, used to test
the host Arnoldi / Krylov-Schur logic (nekstab_amd/krylov.py) without a GPU."""
import numpy as np


class Vec:
    def __init__(self, n):
        self.a = np.zeros(n)


class DenseBackend:
    def __init__(self, A, w=None):
        self.A = A
        self.n = A.shape[0]
        self.w = np.ones(self.n) if w is None else w
        self.nmat = 0

    def alloc(self, n=1):
        return [Vec(self.n) for _ in range(n)]

    def matvec(self, f, q, mode=0):
        f.a = (self.A if mode == 0 else self.A.T) @ q.a
        self.nmat += 1

    def matvec_batch(self, fs, qs, mode=0):
        for f, q in zip(fs, qs):
            self.matvec(f, q, mode)

    def dot(self, p, q):
        return float(np.sum(p.a * self.w * q.a))

    def norm(self, p):
        return np.sqrt(self.dot(p, p))

    def scal(self, p, a):
        p.a = p.a * a

    def copy(self, dst, src):
        dst.a = src.a.copy()

    def orth(self, f, Q):
        h = np.zeros(len(Q))
        for _ in range(2):
            c = np.array([self.dot(f, q) for q in Q])
            for ci, q in zip(c, Q):
                f.a = f.a - ci * q.a
            h += c
        beta = self.norm(f)
        f.a = f.a / beta
        return h, beta

    def basis_gemm(self, Q, Z):
        M = np.stack([q.a for q in Q], axis=1) @ Z
        for i, q in enumerate(Q):
            q.a = M[:, i].copy()

    def basis_gemv(self, Q, y, re, im=None):
        M = np.stack([q.a for q in Q], axis=1)
        re.a = M @ np.real(y)
        if im is not None:
            im.a = M @ np.imag(y)
