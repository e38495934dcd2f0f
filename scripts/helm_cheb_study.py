"""CPU study (oracle operators): would a reduction-free velocity solve pay?  Jacobi-preconditioned CG (what
the GPU runs: three phases per iteration, two of them reductions) against a fixed-degree Chebyshev-Jacobi iteration (no dot
products, spectrum bounds of D^-1 H from a Lanczos run at set-up) on the cylinder's Helmholtz operator H = nu A + (11/6)/dt B at
the production tolerance 3e-12 and at 1e-6.  Prints the iterations both need for the same residual reduction."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from oracle.linns import LinNS2D
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
o = LinNS2D(x=c.x, y=c.y, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re, endtime=c.endtime,
            lxd=c.lxd, has_outflow=c.has_outflow, build_solvers=False)
n, nel = o.n, o.nel
h1, h2 = o.nu, (11.0 / 6.0) / o.dt
gm = o.gmask
def Hop(ug):
    w = o.axhelm(ug[o.gid], h1, h2)
    return gm * np.bincount(o.gflat, weights=w.ravel(), minlength=o.nglob)
K = o._local_matrices(lambda u: o.axhelm(u, h1, h2), n)
diag = np.bincount(o.gflat, weights=np.einsum("eii->ei", K).ravel(), minlength=o.nglob)
Minv = lambda r: gm * r / diag
def pcg(b, tol, maxit=400):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z
    bn = np.sqrt(b @ b)
    for it in range(1, maxit + 1):
        Ap = Hop(p); a = rz / (p @ Ap); x += a * p; r -= a * Ap
        if np.sqrt(r @ r) <= tol * bn: return it
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit
# spectrum bounds of D^-1 H: Lanczos on D^-1/2 H D^-1/2
rng = np.random.default_rng(0)
sd = np.sqrt(1.0 / np.where(gm > 0, diag, 1.0)) * gm
q = gm * rng.standard_normal(o.nglob); q /= np.linalg.norm(q)
qp = np.zeros_like(q); al, be = [], []; bprev = 0.0
for it in range(60):
    w = sd * Hop(sd * q) - bprev * qp
    a = w @ q; w -= a * q; b = np.linalg.norm(w)
    al.append(a); be.append(b)
    if b < 1e-14: break
    qp, q, bprev = q, w / b, b
T = np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1)
ev = np.linalg.eigvalsh(T)
lmin, lmax = ev[0], ev[-1]
print("lx1 %d: spectrum of D^-1 H in [%.4f, %.4f] (Lanczos, 60 steps), condition %.2f" % (lx1, lmin, lmax, lmax / lmin))
def cheb(b, tol, lo, hi, maxit=400):
    th, de = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sig = th / de
    x = np.zeros_like(b); r = b.copy(); bn = np.sqrt(b @ b)
    rho = 1.0 / sig; d = Minv(r) / th
    for it in range(1, maxit + 1):
        x += d; r = r - Hop(d)
        if np.sqrt(r @ r) <= tol * bn: return it
        rho2 = 1.0 / (2.0 * sig - rho)
        d = rho2 * rho * d + 2.0 * rho2 / de * Minv(r)
        rho = rho2
    return maxit
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
smooth = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1)[0]
print("| right-hand side | tolerance | CG iterations | Chebyshev, bounds x (0.95, 1.05) | Chebyshev, exact Lanczos bounds |")
print("|---|---|---|---|---|")
for name, f in (("noise", rng.standard_normal(c.x.shape)), ("leading eigenmode", smooth)):
    b = gm * np.bincount(o.gflat, weights=(o.bm1 * f * c.mask).ravel(), minlength=o.nglob)
    for tol in (1e-6, 3e-12):
        print("| %s | %g | %d | %d | %d |" % (name, tol, pcg(b, tol), cheb(b, tol, 0.95 * lmin, 1.05 * lmax), cheb(b, tol, lmin, lmax)), flush=True)
