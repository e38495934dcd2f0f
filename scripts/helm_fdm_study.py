#!/usr/bin/env python3
"""Velocity Helmholtz preconditioner on stretched boxes (config 5's family): PCG iteration counts with the reference's Jacobi
diagonal and with an element-block fast-diagonalisation (FDM) preconditioner, on the CPU with the 3-D oracle's operators.

  M_fdm^-1 = sum_e R_e^T W (S x S x S) (h2 + nu (lx + ly + lz))^-1 (S x S x S)^T W R_e ,   W = multiplicity^-1/2,
  1-D pairs (A_d, B_d) of the element's GLL grid scaled by its length in direction d, Dirichlet ends where the face is a wall.

    python scripts/helm_fdm_study.py [nelx=8] [lx1=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.linalg as sla
from nekstab_amd import mesh3d
from oracle.linns3d import LinNS3D
from oracle.linns import zwgll, deriv_mat
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stretch = lambda xi: 0.5 * (1.0 - np.cos(np.pi * xi))
c = mesh3d.box_case_3d(ne, ne, ne, n, lengths=(1.0, 1.0, 1.0), re=1000.0, endtime=0.02, stretch=stretch)
sx, sy, sz = np.sin(np.pi * c.x), np.sin(np.pi * c.y), np.sin(np.pi * c.z)
c.ub[0] = sx ** 2 * np.sin(2 * np.pi * c.y) * sz ** 2 * c.mask
c.ub[1] = -np.sin(2 * np.pi * c.x) * sy ** 2 * sz ** 2 * c.mask
o = LinNS3D(x=c.x, y=c.y, z=c.z, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re, endtime=c.endtime, lxd=c.lxd, has_outflow=c.has_outflow, build_solvers=False)
h1, h2 = o.nu, (11.0 / 6.0) / o.dt
print("box %d^3, lx1 %d, dt %.3e, nsteps %d, h2 %.3e, nu %.1e, cell ratio %.1f" % (ne, n, o.dt, o.nsteps, h2, h1, np.diff(stretch(np.linspace(0, 1, ne + 1))).max() / np.diff(stretch(np.linspace(0, 1, ne + 1))).min()))
mask, mult = c.mask, o.dssum(np.ones_like(c.x))
Hop = lambda u: o.dssum(o.axhelm(u, h1, h2)) * mask
# Jacobi diagonal (assembled)
z1, w1 = zwgll(n); D = deriv_mat(z1)
e = np.zeros_like(c.x)
dA = np.zeros_like(c.x)
# diagonal of axhelm by probing unit vectors per local node is too slow: use the formula of undeformed boxes
Lx = c.x[:, 0, 0, -1] - c.x[:, 0, 0, 0]; Ly = c.y[:, 0, -1, 0] - c.y[:, 0, 0, 0]; Lz = c.z[:, -1, 0, 0] - c.z[:, 0, 0, 0]
A1 = D.T @ np.diag(w1) @ D                      # 1-D stiffness on [-1, 1]
B1 = np.diag(w1)
dgA = np.diag(A1); dgB = w1
diagH = np.zeros_like(c.x)
for el in range(c.nel):
    ax_, ay_, az_ = (2.0 / Lx[el]) * dgA, (2.0 / Ly[el]) * dgA, (2.0 / Lz[el]) * dgA
    bx_, by_, bz_ = (Lx[el] / 2) * dgB, (Ly[el] / 2) * dgB, (Lz[el] / 2) * dgB
    diagH[el] = h1 * (az_[:, None, None] * by_[None, :, None] * bx_[None, None, :] + bz_[:, None, None] * ay_[None, :, None] * bx_[None, None, :] + bz_[:, None, None] * by_[None, :, None] * ax_[None, None, :]) \
        + h2 * bz_[:, None, None] * by_[None, :, None] * bx_[None, None, :]
dinv = mask / o.dssum(diagH)
jac = lambda r: dinv * r
# element FDM blocks
def pair(L, dl, dr):
    A = (2.0 / L) * A1; B = (L / 2) * B1
    idx = np.arange(n)[(1 if dl else 0):(n - 1 if dr else n)]
    lam, S = sla.eigh(A[np.ix_(idx, idx)], B[np.ix_(idx, idx)])
    Sf = np.zeros((n, len(idx))); Sf[idx] = S
    return Sf, lam
fd = []
for el in range(c.nel):
    m = c.mask[el]
    dirs = []
    for ax_i, L in ((2, Lx[el]), (1, Ly[el]), (0, Lz[el])):
        lo = np.take(m, 0, axis=ax_i).max() == 0.0; hi = np.take(m, -1, axis=ax_i).max() == 0.0
        dirs.append(pair(L, lo, hi))
    fd.append(dirs)
W = mask / np.sqrt(mult)
def fdm(r):
    out = np.zeros_like(r)
    rw = r * W
    for el in range(c.nel):
        (Sx, lx), (Sy, ly), (Sz, lz) = fd[el]
        t = np.einsum("kji,ia->kja", rw[el], Sx); t = np.einsum("kja,jb->kba", t, Sy); t = np.einsum("kba,kc->cba", t, Sz)
        t = t / (h2 + h1 * (lz[:, None, None] + ly[None, :, None] + lx[None, None, :]))
        t = np.einsum("cba,kc->kba", t, Sz); t = np.einsum("kba,jb->kja", t, Sy); t = np.einsum("kja,ia->kji", t, Sx)
        out[el] = t
    return o.dssum(out * W) * mask
def pcg(b, prec, tol, maxit=600):
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy()
    dot = lambda a, bb: float(np.sum(a * bb / mult))
    rz = dot(r, z); b0 = np.sqrt(dot(b, b))
    for it in range(1, maxit + 1):
        w = Hop(p); al = rz / dot(p, w); x += al * p; r -= al * w
        if np.sqrt(dot(r, r)) <= tol * b0:
            return x, it
        z = prec(r); rz2 = dot(r, z); p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit
rng = np.random.default_rng(0)
for name, f in (("smooth", np.sin(2 * np.pi * c.x) * np.sin(3 * np.pi * c.y) * np.sin(2 * np.pi * c.z)), ("noise (C0)", o.dssum(rng.standard_normal(c.x.shape)) / mult)):
    b = o.dssum(o.bm1 * f) * mask
    for tol in (1e-6, 1e-9):
        _, ij = pcg(b, jac, tol); xf, ifd = pcg(b, fdm, tol)
        print("%-11s tol %.0e: Jacobi %3d iterations, element FDM blocks %3d" % (name, tol, ij, ifd), flush=True)
