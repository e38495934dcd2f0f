"""Case preparation on the host: everything Nek5000 does once at start-up that
the hot path consumes as plain arrays.

 * GLL-node geometry from ``.re2`` vertices + circular-arc sides  [UPSTREAM genxyz/arcsrf]
 * C0 global numbering from ``.ma2`` vertex ids (or from coordinates)
 * velocity Dirichlet masks from the boundary-condition codes
 * nekStab's sponge profile and the sponge-masked inner-product weight
   (reference: core/utils.f:205-342, core/usr_extra.f:102-118)
 * base-flow interpolation between polynomial orders (what Nek's ``load_fld``
   does when the file's lx1 differs from SIZE; used by the reference's own
   lx1=8 adjoint run on the lx1=6 ``BF_1cyl0.f00001``).

The result is a ``Case`` of numpy arrays laid out Nek-style
``(nel, ny, nx)`` (i fastest), ready for ``nsk_init`` (include/nekstab_hip.h).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import nekio
from .quadrature import gauss_lobatto_legendre, interp_matrix

# re2 (counter-clockwise) corner c -> (r,s) sign; faces in preprocessor order:
# face0: s=-1 (c0->c1), face1: r=+1 (c1->c2), face2: s=+1 (c2->c3), face3: r=-1 (c3->c0)
_CORNER_RS = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float)
# ma2 lists vertices lexicographically in (r,s): (c0, c1, c3, c2)
_LEX2CCW = np.array([0, 1, 3, 2])


@dataclass
class Case:
    ndim: int
    nel: int
    lx1: int
    x: np.ndarray            # (nel, lx1, lx1) [j, i]
    y: np.ndarray
    gid: np.ndarray          # (nel, lx1, lx1) int64 0-based global node ids
    nglob: int
    mask: np.ndarray         # (nel, lx1, lx1) 1.0 free / 0.0 Dirichlet (all components)
    ub: np.ndarray           # (ndim, nel, lx1, lx1) base flow
    spng: np.ndarray         # (nel, lx1, lx1) sponge function
    re: float                # Reynolds number (nu = 1/re)
    endtime: float           # sampling period T (param(10))
    cfl: float = 0.5
    lxd: int = 0
    has_outflow: bool = True  # False => pressure null space (adjoint: 'O' -> 'v')
    adjoint: bool = False
    meta: dict = field(default_factory=dict)

    @property
    def lx2(self):
        return self.lx1 - 2


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------

def element_coords_2d(mesh: nekio.Re2Mesh, lx1: int):
    """GLL coordinates for every element: bilinear vertex map plus circular-arc
    side perturbations blended linearly to the opposite side (Nek genxyz/arcsrf)."""
    z, _ = gauss_lobatto_legendre(lx1)
    hm, hp = 0.5 * (1.0 - z), 0.5 * (1.0 + z)
    xc, yc = mesh.xc, mesh.yc
    # bilinear: index [e, j, i]
    x = (xc[:, 0, None, None] * hm[None, :, None] * hm[None, None, :]
         + xc[:, 1, None, None] * hm[None, :, None] * hp[None, None, :]
         + xc[:, 2, None, None] * hp[None, :, None] * hp[None, None, :]
         + xc[:, 3, None, None] * hp[None, :, None] * hm[None, None, :])
    y = (yc[:, 0, None, None] * hm[None, :, None] * hm[None, None, :]
         + yc[:, 1, None, None] * hm[None, :, None] * hp[None, None, :]
         + yc[:, 2, None, None] * hp[None, :, None] * hp[None, None, :]
         + yc[:, 3, None, None] * hp[None, :, None] * hm[None, None, :])
    for (e, f, prm, ctype) in mesh.curves:
        if ctype != "C":
            if ctype in ("", " "):
                continue
            raise NotImplementedError(f"curve type {ctype!r}")
        radius = prm[0]
        if radius == 0.0:
            continue
        p1 = np.array([xc[e, f], yc[e, f]])
        p2 = np.array([xc[e, (f + 1) % 4], yc[e, (f + 1) % 4]])
        gap = np.hypot(*(p1 - p2))
        if abs(2.0 * radius) <= gap * 1.00001:
            raise ValueError("arc radius too small for chord")
        xs, ys = p2[1] - p1[1], p1[0] - p2[0]
        xys = np.hypot(xs, ys)
        dth = abs(np.arcsin(0.5 * gap / radius))
        mid = 0.5 * (p1 + p2)
        cen = mid - np.array([xs, ys]) / xys * radius * np.cos(dth)
        th0 = np.arctan2(mid[1] - cen[1], mid[0] - cen[0])
        r = z if radius > 0 else -z
        # perturbation at parametric position z along p1 -> p2
        px = cen[0] + abs(radius) * np.cos(th0 + r * dth) - (hm * p1[0] + hp * p2[0])
        py = cen[1] + abs(radius) * np.sin(th0 + r * dth) - (hm * p1[1] + hp * p2[1])
        if f >= 2:          # faces 2,3 run against increasing r / s
            px, py = px[::-1], py[::-1]
        if f == 0:          # s=-1 : blend with (1-s)/2
            x[e] += hm[:, None] * px[None, :]
            y[e] += hm[:, None] * py[None, :]
        elif f == 2:        # s=+1
            x[e] += hp[:, None] * px[None, :]
            y[e] += hp[:, None] * py[None, :]
        elif f == 1:        # r=+1
            x[e] += px[:, None] * hp[None, :]
            y[e] += py[:, None] * hp[None, :]
        else:               # r=-1
            x[e] += px[:, None] * hm[None, :]
            y[e] += py[:, None] * hm[None, :]
    return x, y


# ----------------------------------------------------------------------------
# global numbering
# ----------------------------------------------------------------------------

def vertex_ids_from_coords(mesh: nekio.Re2Mesh, periodic_pairs=(), tol=1e-4):
    """Vertex ids (lexicographic corner order, 1-based) by coordinate matching;
    ``periodic_pairs`` = list of (axis, lo, hi) identifying hi with lo."""
    pts = np.stack([mesh.xc.ravel(), mesh.yc.ravel()], axis=1)
    for ax, lo, hi in periodic_pairs:
        sel = np.abs(pts[:, ax] - hi) < tol
        pts[sel, ax] = lo
    span = max(np.ptp(pts[:, 0]), np.ptp(pts[:, 1]))
    key = np.round(pts / (tol * span)).astype(np.int64)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    ids = inv.reshape(mesh.nel, 4) + 1
    return ids[:, _LEX2CCW]     # ccw -> lexicographic columns (c0,c1,c3,c2)


def global_numbering_2d(vert_lex: np.ndarray, lx1: int):
    """C0 numbering of GLL nodes from vertex ids: vertices, then edge interiors
    (keyed by the unordered vertex pair, ordered from the smaller id), then
    element interiors. Returns (gid (nel,lx1,lx1) 0-based, nglob)."""
    nel = vert_lex.shape[0]
    n = lx1
    v = vert_lex - 1                          # 0-based; columns: (-,-),(+,-),(-,+),(+,+)
    nvert = int(v.max()) + 1
    gid = np.full((nel, n, n), -1, dtype=np.int64)
    gid[:, 0, 0], gid[:, 0, -1], gid[:, -1, 0], gid[:, -1, -1] = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
    # edges: (va, vb, slice) with va at the low-index end
    edges = [(0, 1, (0, slice(1, n - 1))),        # s=-1
             (2, 3, (n - 1, slice(1, n - 1))),    # s=+1
             (0, 2, (slice(1, n - 1), 0)),        # r=-1
             (1, 3, (slice(1, n - 1), n - 1))]    # r=+1
    nint = n - 2
    pairs = []
    for a, b, _ in edges:
        lo = np.minimum(v[:, a], v[:, b])
        hi = np.maximum(v[:, a], v[:, b])
        pairs.append(lo * nvert + hi)
    allpairs = np.concatenate(pairs)
    uniq, inv = np.unique(allpairs, return_inverse=True)
    inv = inv.reshape(4, nel)
    base = nvert
    k = np.arange(nint)
    for m, (a, b, sl) in enumerate(edges):
        fwd = v[:, a] < v[:, b]              # local direction agrees with lo->hi
        pos = np.where(fwd[:, None], k[None, :], nint - 1 - k[None, :])
        ids = base + inv[m][:, None] * nint + pos
        if isinstance(sl[0], int):
            gid[:, sl[0], sl[1]] = ids
        else:
            gid[:, sl[0], sl[1]] = ids
    base += len(uniq) * nint
    ii = np.arange(nint * nint).reshape(nint, nint)
    gid[:, 1:-1, 1:-1] = base + np.arange(nel)[:, None, None] * nint * nint + ii[None]
    nglob = base + nel * nint * nint
    assert gid.min() >= 0
    return gid, int(nglob)


def dirichlet_mask_2d(mesh: nekio.Re2Mesh, lx1: int, dirichlet_codes=("v", "W", "V")):
    mask = np.ones((mesh.nel, lx1, lx1))
    for (e, f, _prm, code) in mesh.bcs:
        if code.strip() in dirichlet_codes:
            if f == 0:
                mask[e, 0, :] = 0.0
            elif f == 1:
                mask[e, :, -1] = 0.0
            elif f == 2:
                mask[e, -1, :] = 0.0
            else:
                mask[e, :, 0] = 0.0
    return mask


# ----------------------------------------------------------------------------
# nekStab sponge (core/utils.f:205-342) and bm1s mask (core/usr_extra.f:102-118)
# ----------------------------------------------------------------------------

def _stepf(x):
    """mth_stepf, core/utils.f:330-342."""
    x = np.asarray(x, dtype=float)
    out = np.ones_like(x)
    lo = x <= 0.001
    mid = (~lo) & (x <= 0.999)
    out[lo] = 0.0
    xm = x[mid]
    out[mid] = 1.0 / (1.0 + np.exp(1.0 / (xm - 1.0) + 1.0 / xm))
    return out


def sponge_function(coords, lspg, rspg, acc=0.333):
    """spng_set: ``coords`` list of coordinate arrays per dimension, ``lspg``/
    ``rspg`` left/right sponge lengths per dimension."""
    acc = abs(acc)
    fun = np.zeros_like(coords[0])
    for c, L, R in zip(coords, lspg, rspg):
        wl, wr = (1.0 - acc) * L, (1.0 - acc) * R
        dl, dr = acc * L, acc * R
        if not (wl > 0.0 or wr > 0.0):
            continue
        bmin, bmax = c.min(), c.max()
        xxmax, xxmin = bmax - wr, bmin + wl
        xxmax_c, xxmin_c = xxmax + dr, xxmin - dl
        r = np.zeros_like(c)
        m1 = c <= xxmin_c
        m2 = (~m1) & (c < xxmin)
        m3 = (~m1) & (~m2) & (c <= xxmax)
        m4 = (~m1) & (~m2) & (~m3) & (c < xxmax_c)
        m5 = ~(m1 | m2 | m3 | m4)
        r[m1] = 1.0
        if wl > 0:
            r[m2] = _stepf((xxmin - c[m2]) / wl)
        if wr > 0:
            r[m4] = _stepf((c[m4] - xxmax) / wr)
        r[m5] = 1.0
        fun = np.maximum(fun, r)
    return fun


# ----------------------------------------------------------------------------
# field transfer between orders
# ----------------------------------------------------------------------------

def interp_field_2d(f: np.ndarray, lx_to: int) -> np.ndarray:
    """(..., ny, nx) GLL(lx_from) -> GLL(lx_to), element by element."""
    lx_from = f.shape[-1]
    if lx_from == lx_to:
        return f.copy()
    J = interp_matrix(gauss_lobatto_legendre(lx_from)[0], gauss_lobatto_legendre(lx_to)[0])
    return np.einsum("ai,bj,...ij->...ab", J, J, f, optimize=True)


# ----------------------------------------------------------------------------
# assembling a Case
# ----------------------------------------------------------------------------

def build_case_2d(mesh: nekio.Re2Mesh, vlex: np.ndarray, bf_u: np.ndarray, lx1: int, *,
                  adjoint=False, endtime=1.0, re=50.0, xlspg=5.0, xrspg=5.0, spng_str=1.7,
                  cfl=0.5, meta=None) -> Case:
    """Assemble a Case the way the reference example sets it up
    (examples/cylinder/stability/direct/{1cyl.par,1cyl.usr,SIZE}):
    ``bf_u`` = base flow (2, nel, lxb, lxb) at any order (interpolated like load_fld)."""
    x, y = element_coords_2d(mesh, lx1)
    gid, nglob = global_numbering_2d(vlex, lx1)
    codes = ("v", "W", "V", "O") if adjoint else ("v", "W", "V")   # 1cyl.usr:126-132
    mask = dirichlet_mask_2d(mesh, lx1, codes)
    gmask = np.ones(nglob)                 # a node fixed in one copy is fixed in all
    np.minimum.at(gmask, gid.ravel(), mask.ravel())
    mask = gmask[gid]
    ub = interp_field_2d(np.asarray(bf_u, dtype=np.float64), lx1)
    if spng_str != 0.0:
        spng = sponge_function([x, y], [xlspg, 0.0], [xrspg, 0.0])
    else:
        spng = np.zeros_like(x)
    has_out = any(b[3].strip() == "O" for b in mesh.bcs) and not adjoint
    m = dict(meta or {})
    m["vert"] = vlex - 1
    m["nvert"] = int(vlex.max())
    return Case(ndim=2, nel=mesh.nel, lx1=lx1, x=x, y=y, gid=gid, nglob=nglob, mask=mask,
                ub=ub, spng=spng, re=re, endtime=endtime, cfl=cfl, lxd=3 * lx1 // 2,
                has_outflow=has_out, adjoint=adjoint, meta=m)


def load_cylinder_case(casedir: str, lx1: int, *, session="1cyl", use_ma2=True, **kw) -> Case:
    """Read ``<session>.re2/.ma2`` and ``BF_<session>0.f00001`` from a nekStab case directory."""
    import os
    mesh = nekio.read_re2(os.path.join(casedir, session + ".re2"))
    if use_ma2:
        vlex, _ = nekio.read_ma2(os.path.join(casedir, session + ".ma2"))
    else:
        vlex = vertex_ids_from_coords(mesh, periodic_pairs=[(1, mesh.yc.min(), mesh.yc.max())])
    bf = nekio.read_fld(os.path.join(casedir, "BF_%s0.f00001" % session))
    return build_case_2d(mesh, vlex, bf.u[:, :, 0], lx1,
                         meta={"casedir": casedir, "session": session, "bf_lx1": bf.nx}, **kw)


def save_case_npz(path: str, mesh: nekio.Re2Mesh, vlex: np.ndarray, bf_u: np.ndarray, bf_p: np.ndarray | None = None):
    """Compact fixture of a case's *data* (mesh vertices, curves, BCs, vertex ids, base flow)."""
    cur = np.array([[c[0], c[1]] + list(c[2]) for c in mesh.curves if c[3] == "C"], dtype=np.float64)
    bce = np.array([[b[0], b[1]] for b in mesh.bcs], dtype=np.int32)
    bcc = np.array([b[3] for b in mesh.bcs])
    np.savez_compressed(path, xc=mesh.xc, yc=mesh.yc, curves=cur, bc_ef=bce, bc_code=bcc,
                        vlex=vlex.astype(np.int32), bf_u=bf_u, **({} if bf_p is None else {"bf_p": bf_p}))


def load_case_npz(path: str, lx1: int, **kw) -> Case:
    z = np.load(path)
    curves = [(int(r[0]), int(r[1]), r[2:7].copy(), "C") for r in z["curves"]]
    bcs = [(int(ef[0]), int(ef[1]), np.zeros(5), str(cd)) for ef, cd in zip(z["bc_ef"], z["bc_code"])]
    mesh = nekio.Re2Mesh(2, z["xc"].shape[0], z["xc"], z["yc"], None, curves, bcs)
    meta = {"npz": path}
    if "bf_p" in z.files:                      # base-flow pressure on mesh 1 (as in the field file), at the file's order
        meta["bf_p"] = z["bf_p"].astype(np.float64)
    return build_case_2d(mesh, z["vlex"].astype(np.int64), z["bf_u"].astype(np.float64), lx1, meta=meta, **kw)


# ----------------------------------------------------------------------------
# uniform 2x2 refinement of a Case (BASELINE config 3: E ~ 8k)
# ----------------------------------------------------------------------------

def refine_case_2x2(case: Case) -> Case:
    """Split every element into 2x2 children.  Child geometry / base flow are the parent's
    polynomials evaluated on the children's GLL nodes (exactly conforming); vertex ids by
    coordinate matching (periodic in y), Dirichlet faces inherited."""
    from scipy.spatial import cKDTree
    n = case.lx1
    z = gauss_lobatto_legendre(n)[0]
    Jh = [interp_matrix(z, 0.5 * (z - 1.0)), interp_matrix(z, 0.5 * (z + 1.0))]     # lower / upper half
    nel = case.nel

    def split(f):                      # (..., nel, n, n) -> (..., 4 nel, n, n), child order (qs, qr)
        out = []
        for qs in range(2):
            for qr in range(2):
                out.append(np.einsum("ai,bj,...ij->...ab", Jh[qs], Jh[qr], f, optimize=True))
        o = np.stack(out, axis=-3)     # (..., nel, 4, n, n)
        return o.reshape(f.shape[:-3] + (4 * nel, n, n))

    x, y, ub = split(case.x), split(case.y), split(case.ub)
    # face flags of the parent from its nodal mask (interior face nodes all fixed)
    pm = case.mask
    f_sm = np.all(pm[:, 0, 1:-1] == 0, axis=1); f_sp = np.all(pm[:, -1, 1:-1] == 0, axis=1)
    f_rm = np.all(pm[:, 1:-1, 0] == 0, axis=1); f_rp = np.all(pm[:, 1:-1, -1] == 0, axis=1)
    mask = np.ones((nel, 4, n, n))
    for qs in range(2):
        for qr in range(2):
            c = qs * 2 + qr
            if qs == 0: mask[f_sm, c, 0, :] = 0.0
            if qs == 1: mask[f_sp, c, -1, :] = 0.0
            if qr == 0: mask[f_rm, c, :, 0] = 0.0
            if qr == 1: mask[f_rp, c, :, -1] = 0.0
    mask = mask.reshape(4 * nel, n, n)
    # vertex ids from corner coordinates, periodic in y
    cx = np.stack([x[:, 0, 0], x[:, 0, -1], x[:, -1, 0], x[:, -1, -1]], axis=1)
    cy = np.stack([y[:, 0, 0], y[:, 0, -1], y[:, -1, 0], y[:, -1, -1]], axis=1)
    ymin, ymax = case.y.min(), case.y.max()
    pts = np.stack([cx.ravel(), np.where(np.abs(cy.ravel() - ymax) < 1e-6, ymin, cy.ravel())], axis=1)
    tree = cKDTree(pts)
    parent = np.arange(len(pts))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for a, b in tree.query_pairs(r=2e-5):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    roots = np.array([find(a) for a in range(len(pts))])
    _, vid = np.unique(roots, return_inverse=True)
    vlex = vid.reshape(4 * nel, 4) + 1
    gid, nglob = global_numbering_2d(vlex, n)
    gmask = np.ones(nglob)
    np.minimum.at(gmask, gid.ravel(), mask.ravel())
    mask = gmask[gid]
    spng = sponge_function([x, y], [5.0, 0.0], [5.0, 0.0]) if case.spng.max() > 0 else np.zeros_like(x)
    m = dict(case.meta)
    m["vert"] = vlex - 1
    m["nvert"] = int(vlex.max())
    return Case(ndim=2, nel=4 * nel, lx1=n, x=x, y=y, gid=gid, nglob=nglob, mask=mask, ub=ub, spng=spng,
                re=case.re, endtime=case.endtime, cfl=case.cfl, lxd=case.lxd, has_outflow=case.has_outflow,
                adjoint=case.adjoint, meta=m)
