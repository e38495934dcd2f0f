"""CPU study (oracle E matrix, cylinder mesh): right-preconditioned GMRES iteration counts of the pressure solve with
restricted additive Schwarz patches + vertex coarse space combined (a) additively (what the GPU path does),
(b) multiplicatively, coarse first: z = z_c + M_loc (r - E z_c)."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from oracle.linns import LinNS2D
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
t0 = time.time()
o = LinNS2D(x=c.x, y=c.y, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re, endtime=c.endtime,
            lxd=c.lxd, has_outflow=c.has_outflow)
E = o._Emat.tocsr()
n, m, nel = o.n, o.m, o.nel
MM = m * m
print("E built %.1fs, npr %d nnz %d" % (time.time() - t0, o.npr, E.nnz), flush=True)
# neighbours through shared velocity nodes
owner = {}
for e in range(nel):
    for g in np.unique(c.gid[e]): owner.setdefault(int(g), []).append(e)
patches = []
for e in range(nel):
    dof = [e * MM + k for k in range(MM)]
    ge = set(np.unique(c.gid[e]).tolist())
    nbs = sorted({f for g in ge for f in owner[g] if f != e})
    for f in nbs:
        sh = np.isin(c.gid[f], list(ge))
        jj, ii = np.where(sh)
        jmin, jmax, imin, imax = jj.min(), jj.max(), ii.min(), ii.max()
        b0, b1, a0, a1 = 0, m, 0, m
        if jmin == jmax:
            if jmin == 0: b1 = L
            elif jmin == n - 1: b0 = m - L
        if imin == imax:
            if imin == 0: a1 = L
            elif imin == n - 1: a0 = m - L
        if (b0, b1, a0, a1) == (0, m, 0, m): continue
        dof += [f * MM + b * m + a for b in range(b0, b1) for a in range(a0, a1)]
    patches.append(np.array(dof))
print("patch sizes", np.bincount([len(p) for p in patches]).nonzero()[0], flush=True)
Pinv = []
Ec = E.tocsc()
for e in range(nel):
    p = patches[e]
    A = E[p][:, p].toarray()
    Pinv.append(np.linalg.inv(A)[:MM, :])          # restricted: own rows only
# coarse space: bilinear hats at the Gauss points
z2 = o.z2
hm, hp = 0.5 * (1 - z2), 0.5 * (1 + z2)
hat = np.stack([np.outer(hm, hm), np.outer(hm, hp), np.outer(hp, hm), np.outer(hp, hp)]).reshape(4, MM)   # [vertex (s,r)] x [b,a]
vert = np.asarray(c.meta["vert"]); nv = int(c.meta["nvert"])
rows = (np.arange(nel)[:, None, None] * MM + np.arange(MM)[None, None, :]).repeat(4, 1).ravel()
cols = vert[:, :, None].repeat(MM, 2).ravel()
R = sp.coo_matrix((np.tile(hat, (nel, 1, 1)).ravel(), (rows, cols)), shape=(o.npr, nv)).tocsr()
Ac = (R.T @ E @ R).tocsc()
Aclu = spla.splu(Ac)
EP = (E @ R).tocsr()
def M_loc(r):
    z = np.empty(o.npr)
    for e in range(nel): z[e * MM:(e + 1) * MM] = Pinv[e] @ r[patches[e]]
    return z
def M_add(r): return M_loc(r) + R @ Aclu.solve(R.T @ r)
def M_mul(r):
    xc = Aclu.solve(R.T @ r)
    return R @ xc + M_loc(r - EP @ xc)
def M_mul2(r):                                     # coarse - local - coarse (symmetrised)
    xc = Aclu.solve(R.T @ r)
    z = R @ xc + M_loc(r - EP @ xc)
    return z + R @ Aclu.solve(R.T @ (r - E @ z))
def gmres_counts(b, M, tols, maxit=60):
    # right-preconditioned GMRES, modified Gram-Schmidt, true residual norm from the Givens recurrence
    V = [b / np.linalg.norm(b)]; H = np.zeros((maxit + 1, maxit)); g = np.zeros(maxit + 1); g[0] = np.linalg.norm(b)
    cs, sn = np.zeros(maxit), np.zeros(maxit); out = {}; b0 = g[0]
    for j in range(maxit):
        w = E @ M(V[j])
        for i in range(j + 1): H[i, j] = w @ V[i]; w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w); V.append(w / H[j + 1, j])
        for i in range(j):
            t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]; H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]; H[i, j] = t
        rho = np.hypot(H[j, j], H[j + 1, j]); cs[j], sn[j] = H[j, j] / rho, H[j + 1, j] / rho
        H[j, j] = rho; g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j]
        for t in tols:
            if t not in out and abs(g[j + 1]) <= t * b0: out[t] = j + 1
        if len(out) == len(tols): break
    return out
rng = np.random.default_rng(0)
from nekstab_amd import seed
qx, qy = seed.add_noise(c)
rhs = {"noise": rng.standard_normal(o.npr), "div(noise velocity)": o.opdiv(qx, qy).ravel()}
tols = (1e-1, 1e-2, 1e-3, 1e-4)
for name, b in rhs.items():
    for pname, M in (("additive (current)", M_add), ("coarse-first multiplicative", M_mul), ("coarse-local-coarse", M_mul2)):
        t0 = time.time(); r = gmres_counts(b, M, tols)
        print("%-20s %-28s %s  (%.0fs)" % (name, pname, " ".join("%g:%s" % (t, r.get(t, ">60")) for t in tols), time.time() - t0), flush=True)

if os.environ.get("SPECTRUM"):
    A = spla.LinearOperator((o.npr, o.npr), matvec=lambda v: E @ M_add(v))
    big = spla.eigs(A, k=6, which="LM", return_eigenvectors=False, tol=1e-3)
    print("largest |eig| of E M^-1:", np.sort(np.abs(big))[::-1][:6])
    # smallest via a few hundred Arnoldi steps on the inverse-free operator: use GMRES Hessenberg Ritz values instead
    V = [rhs["noise"] / np.linalg.norm(rhs["noise"])]; kk = 80; H = np.zeros((kk + 1, kk))
    for j in range(kk):
        w = A.matvec(V[j])
        for i in range(j + 1): H[i, j] = w @ V[i]; w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w); V.append(w / H[j + 1, j])
    ev = np.linalg.eigvals(H[:kk, :kk])
    print("Ritz values (80 steps): min real %.4f, max real %.4f; smallest 8 |.|: %s" % (ev.real.min(), ev.real.max(), np.sort(np.abs(ev))[:8]))
    for om in (0.5, 2.0, 4.0):
        Mw = lambda r, om=om: M_loc(r) + om * (R @ Aclu.solve(R.T @ r))
        print("coarse weight", om, gmres_counts(rhs["noise"], Mw, tols))
    # local part alone and coarse alone, to see who limits
    print("local only", gmres_counts(rhs["noise"], M_loc, tols))

if os.environ.get("MIDLEVEL"):
    # enriched coarse spaces (exact Galerkin solves): continuous piecewise-bilinear functions on every element cut
    # into s x s cells (s = 2, 3), i.e. the vertex space refined
    for s_ in (2, 3):
        # node coordinates in the reference element: k/s ; global numbering through rounded physical coordinates
        tt = np.linspace(-1, 1, s_ + 1)
        # 1-D hats on the refined grid evaluated at the Gauss points
        hh = np.zeros((s_ + 1, m))
        for k in range(s_ + 1):
            e_k = np.zeros(s_ + 1); e_k[k] = 1.0
            hh[k] = np.interp(z2, tt, e_k)
        # physical coordinates of the refined nodes (bilinear map of the element vertices is enough for numbering)
        from oracle.linns import interp_mat
        Jv = interp_mat(o.z1, tt)
        xn = np.einsum("ja,eab,ib->eji", Jv, c.x, Jv); yn = np.einsum("ja,eab,ib->eji", Jv, c.y, Jv)
        key = np.round(np.stack([xn, yn], -1) * 1e6).astype(np.int64).reshape(-1, 2)
        _, gidn = np.unique(key, axis=0, return_inverse=True)
        gidn = gidn.reshape(nel, s_ + 1, s_ + 1); nn_ = gidn.max() + 1
        vals = np.einsum("jb,ia->jiba", hh, hh).reshape((s_ + 1) ** 2, MM)            # [node (j,i)] x [gauss (b,a)]
        rows = (np.arange(nel)[:, None, None] * MM + np.arange(MM)[None, None, :]).repeat((s_ + 1) ** 2, 1).ravel()
        cols = gidn.reshape(nel, -1)[:, :, None].repeat(MM, 2).ravel()
        R2 = sp.coo_matrix((np.tile(vals, (nel, 1, 1)).ravel(), (rows, cols)), shape=(o.npr, nn_)).tocsr()
        A2 = (R2.T @ E @ R2).tocsc()
        lu2 = spla.splu(A2)
        M2 = lambda r: M_loc(r) + R2 @ lu2.solve(R2.T @ r)
        for name, b in rhs.items():
            print("refined vertex space s=%d (%d dofs) + patches: %-20s %s" % (s_, nn_, name, gmres_counts(b, M2, tols)), flush=True)
