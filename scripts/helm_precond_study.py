"""CPU study (oracle operators): PCG iteration counts of the velocity Helmholtz solve on the cylinder
mesh with (a) the Jacobi preconditioner of the reference's solver, (b) an element-block
Neumann-Neumann preconditioner (exact local inverses, 1/multiplicity weights), (c) interior blocks +
Jacobi on the interface nodes."""
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
print("lx1", lx1, "dt", o.dt, "h2", h2)
gm = o.gmask
def Hop(ug):
    w = o.axhelm(ug[o.gid], h1, h2)
    return gm * np.bincount(o.gflat, weights=w.ravel(), minlength=o.nglob)
K = o._local_matrices(lambda u: o.axhelm(u, h1, h2), n)            # (nel, n^2, n^2)
diag = np.bincount(o.gflat, weights=np.einsum("eii->ei", K).ravel(), minlength=o.nglob)
mult = np.bincount(o.gflat, minlength=o.nglob).astype(float)
# (b) Neumann-Neumann: local inverse of the masked element matrix
mk_l = gm[o.gid].reshape(nel, n * n)
Kinv = np.empty_like(K)
for e in range(nel):
    Ke = K[e] * mk_l[e][:, None] * mk_l[e][None, :] + np.diag(1.0 - mk_l[e])
    Kinv[e] = np.linalg.inv(Ke)
wloc = (1.0 / mult)[o.gid].reshape(nel, n * n)
def M_nn(rg):
    rl = rg[o.gid].reshape(nel, n * n) * wloc
    zl = np.einsum("eij,ej->ei", Kinv, rl) * wloc * mk_l
    return gm * np.bincount(o.gflat, weights=zl.ravel(), minlength=o.nglob)
# (c) interior blocks + Jacobi interface
ii = np.zeros((n, n), bool); ii[1:-1, 1:-1] = True
iidx = np.where(ii.ravel())[0]
Kii_inv = np.stack([np.linalg.inv(K[e][np.ix_(iidx, iidx)]) for e in range(nel)])
def M_int(rg):
    z = rg / diag
    rl = rg[o.gid].reshape(nel, n * n)[:, iidx]
    zl = np.einsum("eij,ej->ei", Kii_inv, rl)
    z[o.gid.reshape(nel, n * n)[:, iidx].ravel()] = zl.ravel()
    return gm * z
def pcg(b, M, tol=1e-10, maxit=400):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z
    bn = np.sqrt(b @ b)
    for it in range(1, maxit + 1):
        Ap = Hop(p); a = rz / (p @ Ap); x += a * p; r -= a * Ap
        if np.sqrt(r @ r) <= tol * bn: return it, x
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit, x
rng = np.random.default_rng(0)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
smooth = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1)[0]
for name, f in (("noise", rng.standard_normal(c.x.shape)), ("eigenmode", smooth)):
    b = gm * np.bincount(o.gflat, weights=(o.bm1 * f * c.mask).ravel(), minlength=o.nglob)
    for pname, M in (("jacobi", lambda r: gm * r / diag), ("neumann-neumann", M_nn), ("interior+jacobi", M_int)):
        for tol in (1e-4, 1e-10):
            it, _ = pcg(b, M, tol)
            print("%-10s %-16s tol %.0e: %d iterations" % (name, pname, tol, it), flush=True)

# (d) the same Neumann-Neumann preconditioner with fast-diagonalisation local solves: separable
# approximation K_e ~ h2 Bs x Br + nu (Bs x Ar + As x Br) from the element's mean extents, Neumann ends
# everywhere, Dirichlet nodes handled by masking the result only
import scipy.linalg as sla
w1 = o.w1; Ah = o.D.T @ np.diag(w1) @ o.D; Bh = np.diag(w1)
lam, S = sla.eigh(Ah, Bh)                     # S^T Bh S = I, S^T Ah S = lam
x, y = c.x, c.y
Lr = np.sqrt((x[:, :, -1] - x[:, :, 0]) ** 2 + (y[:, :, -1] - y[:, :, 0]) ** 2)
Ls = np.sqrt((x[:, -1, :] - x[:, 0, :]) ** 2 + (y[:, -1, :] - y[:, 0, :]) ** 2)
Lr = (Lr * w1[None, :]).sum(1) / 2.0; Ls = (Ls * w1[None, :]).sum(1) / 2.0      # weighted mean extents
def M_fdm(rg):
    rl = rg[o.gid] * wloc.reshape(nel, n, n)
    # local solve: z = (Ss x Sr) [ (Lr Ls /4) h2 + nu (Ls/Lr lam_r + Lr/Ls lam_s) ]^-1 (Ss x Sr)^T r
    t = np.einsum("ja,eji,ib->eab", S, rl, S)
    den = h2 * (Lr * Ls / 4.0)[:, None, None] + h1 * ((Ls / Lr)[:, None, None] * lam[None, None, :] + (Lr / Ls)[:, None, None] * lam[None, :, None])
    t = t / den
    zl = np.einsum("ja,eab,ib->eji", S, t, S) * wloc.reshape(nel, n, n) * mk_l.reshape(nel, n, n)
    return gm * np.bincount(o.gflat, weights=zl.ravel(), minlength=o.nglob)
for name, f in (("noise", rng.standard_normal(c.x.shape)), ("eigenmode", smooth)):
    b = gm * np.bincount(o.gflat, weights=(o.bm1 * f * c.mask).ravel(), minlength=o.nglob)
    for tol in (1e-4, 1e-6, 1e-10):
        print("%-10s %-16s tol %.0e: %d iterations   (jacobi %d, exact NN %d)" % (name, "NN-FDM", tol, pcg(b, M_fdm, tol)[0], pcg(b, lambda r: gm * r / diag, tol)[0], pcg(b, M_nn, tol)[0]), flush=True)
