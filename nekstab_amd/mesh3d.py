"""3-D cases for the hot path (BASELINE configs 4 and 5 are 3-D): extrusion of a 2-D case in z
(what the reference's workflow does with ``n2to3`` for the backward-facing step, periodic in z)
and structured hexahedral boxes (``genbox`` for the lid-driven cavity, examples/lid_driven/cav.box).

Arrays are Nek-style ``(nel, lz1, ly1, lx1)`` = ``[e, k, j, i]`` (i fastest); the element-corner
vertex table is lexicographic in (r, s, t): column c = ir + 2*is + 4*it.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .mesh import Case
from .quadrature import gauss_lobatto_legendre


@dataclass
class Case3D:
    ndim: int
    nel: int
    lx1: int
    x: np.ndarray            # (nel, lx1, lx1, lx1) [k, j, i]
    y: np.ndarray
    z: np.ndarray
    gid: np.ndarray          # int64 0-based global node ids
    nglob: int
    mask: np.ndarray         # 1.0 free / 0.0 Dirichlet (all components)
    ub: np.ndarray           # (3, nel, lx1, lx1, lx1) base flow
    spng: np.ndarray
    re: float
    endtime: float
    cfl: float = 0.5
    lxd: int = 0
    has_outflow: bool = True
    adjoint: bool = False
    meta: dict = field(default_factory=dict)

    @property
    def lx2(self):
        return self.lx1 - 2


def extrude_case(c2: Case, nz: int, lz: float, *, periodic=True, z0=0.0) -> Case3D:
    """nz uniform layers of height lz/nz; element e3 = kz*nel2 + e2.  Periodic in z (spanwise-homogeneous
    stability problems) or with Dirichlet walls at both ends."""
    if periodic and nz < 2:
        raise ValueError("periodic extrusion needs nz >= 2")
    n = c2.lx1
    zg, _ = gauss_lobatto_legendre(n)
    dz = lz / nz
    nel2 = c2.nel
    nel = nel2 * nz
    ext = lambda f: np.broadcast_to(f[None, :, None, :, :], (nz, nel2, n, n, n)).reshape(nel, n, n, n).copy()
    x, y, spng, mask = ext(c2.x), ext(c2.y), ext(c2.spng), ext(c2.mask)
    zl = z0 + dz * (np.arange(nz)[:, None] + 0.5 * (1.0 + zg[None, :]))          # (nz, n)
    z = np.broadcast_to(zl[:, None, :, None, None], (nz, nel2, n, n, n)).reshape(nel, n, n, n).copy()
    nlev = nz * (n - 1) if periodic else nz * (n - 1) + 1
    lev = (np.arange(nz)[:, None] * (n - 1) + np.arange(n)[None, :]) % nlev     # (nz, n)
    gid = (c2.gid[None, :, None, :, :] + c2.nglob * lev[:, None, :, None, None]).reshape(nel, n, n, n)
    if not periodic:
        m5 = mask.reshape(nz, nel2, n, n, n)
        m5[0, :, 0] = 0.0
        m5[-1, :, -1] = 0.0
    ub = np.zeros((3, nel, n, n, n))
    ub[0], ub[1] = ext(c2.ub[0]), ext(c2.ub[1])
    v2 = np.asarray(c2.meta["vert"], dtype=np.int64)                             # (nel2, 4) lexicographic
    nv2 = int(c2.meta["nvert"])
    nvl = nz if periodic else nz + 1
    vert = np.zeros((nz, nel2, 8), dtype=np.int64)
    for kz in range(nz):
        vert[kz, :, :4] = v2 + nv2 * kz
        vert[kz, :, 4:] = v2 + nv2 * ((kz + 1) % nvl)
    meta = dict(c2.meta)
    meta.update(vert=vert.reshape(nel, 8), nvert=nv2 * nvl, nz=nz, lz=lz, periodic_z=periodic, nel2=nel2)
    return Case3D(ndim=3, nel=nel, lx1=n, x=x, y=y, z=z, gid=gid.astype(np.int64), nglob=c2.nglob * nlev, mask=mask,
                  ub=ub, spng=spng, re=c2.re, endtime=c2.endtime, cfl=c2.cfl, lxd=3 * n // 2,
                  has_outflow=c2.has_outflow, adjoint=c2.adjoint, meta=meta)


def extrude_field(f2: np.ndarray, nz: int) -> np.ndarray:
    """(nel2, n, n) -> (nz*nel2, n, n, n), constant in z (same element order as ``extrude_case``)."""
    nel2, n = f2.shape[0], f2.shape[-1]
    return np.broadcast_to(f2[None, :, None, :, :], (nz, nel2, n, n, n)).reshape(nz * nel2, n, n, n).copy()


def extrude_pressure(p2: np.ndarray, nz: int) -> np.ndarray:
    nel2, m = p2.shape[0], p2.shape[-1]
    return np.broadcast_to(p2[None, :, None, :, :], (nz, nel2, m, m, m)).reshape(nz * nel2, m, m, m).copy()


def box_case_3d(nx: int, ny: int, nz: int, lx1: int, *, lengths=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
                periodic=(False, False, False), outflow_xmax=False, re=100.0, endtime=0.1, cfl=0.5,
                ub_func=None, warp=0.0, stretch=None) -> Case3D:
    """Structured box of nx*ny*nz hexahedra.  Every non-periodic face is a velocity-Dirichlet wall except
    x = xmax when ``outflow_xmax``.  ``warp`` > 0 deforms the mesh smoothly (all metric terms become non-zero);
    ``stretch`` = callable (xi in [0,1]) -> [0,1] for wall clustering (cav.box uses a Chebyshev-like map)."""
    n = lx1
    zg, _ = gauss_lobatto_legendre(n)
    ne = (nx, ny, nz)
    nel = nx * ny * nz

    def axis_nodes(a):
        edges = np.linspace(0.0, 1.0, ne[a] + 1)
        if stretch is not None:
            edges = stretch(edges)
        lo, hi = edges[:-1], edges[1:]
        return origin[a] + lengths[a] * (lo[:, None] + 0.5 * (hi - lo)[:, None] * (1.0 + zg[None, :]))   # (ne, n)

    xa, ya, za = axis_nodes(0), axis_nodes(1), axis_nodes(2)
    ez, ey, ex = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ex, ey, ez = ex.ravel(), ey.ravel(), ez.ravel()                               # element e = ex + nx*(ey + ny*ez)
    x = np.broadcast_to(xa[ex][:, None, None, :], (nel, n, n, n)).copy()
    y = np.broadcast_to(ya[ey][:, None, :, None], (nel, n, n, n)).copy()
    z = np.broadcast_to(za[ez][:, :, None, None], (nel, n, n, n)).copy()
    ng = [ne[a] * (n - 1) + (0 if periodic[a] else 1) for a in range(3)]
    I = (ex[:, None] * (n - 1) + np.arange(n)[None, :]) % ng[0]
    J = (ey[:, None] * (n - 1) + np.arange(n)[None, :]) % ng[1]
    K = (ez[:, None] * (n - 1) + np.arange(n)[None, :]) % ng[2]
    gid = I[:, None, None, :] + ng[0] * (J[:, None, :, None] + ng[1] * K[:, :, None, None])
    gmask = np.ones(ng)                                                          # [I, J, K]
    if not periodic[0]:
        gmask[0] = 0.0
        if not outflow_xmax:
            gmask[-1] = 0.0
    if not periodic[1]:
        gmask[:, 0] = 0.0; gmask[:, -1] = 0.0
    if not periodic[2]:
        gmask[:, :, 0] = 0.0; gmask[:, :, -1] = 0.0
    mask = gmask.transpose(2, 1, 0).ravel()[gid]                                 # flat index I + ng0*(J + ng1*K)
    if warp:
        lx, ly, lz = lengths
        sx = np.sin(np.pi * (x - origin[0]) / lx) if not periodic[0] else np.sin(2 * np.pi * (x - origin[0]) / lx)
        sy = np.sin(np.pi * (y - origin[1]) / ly) if not periodic[1] else np.sin(2 * np.pi * (y - origin[1]) / ly)
        sz = np.sin(np.pi * (z - origin[2]) / lz) if not periodic[2] else np.sin(2 * np.pi * (z - origin[2]) / lz)
        bump = sx * sy * sz                                                      # vanishes on every wall: faces stay planar
        x, y, z = x + warp * lx * bump * 0.7, y - warp * ly * bump, z + warp * lz * bump * 0.5
    nvg = [ne[a] + (0 if periodic[a] else 1) for a in range(3)]
    vert = np.zeros((nel, 8), dtype=np.int64)
    for c in range(8):
        ir, is_, it = c & 1, (c >> 1) & 1, (c >> 2) & 1
        vert[:, c] = ((ex + ir) % nvg[0]) + nvg[0] * (((ey + is_) % nvg[1]) + nvg[1] * ((ez + it) % nvg[2]))
    ub = np.zeros((3, nel, n, n, n))
    if ub_func is not None:
        ub[:] = ub_func(x, y, z)
    spng = np.zeros_like(x)
    has_out = bool(outflow_xmax) and not periodic[0]
    meta = dict(vert=vert, nvert=int(np.prod(nvg)), box=(nx, ny, nz), periodic=periodic)
    return Case3D(ndim=3, nel=nel, lx1=n, x=x, y=y, z=z, gid=gid.astype(np.int64), nglob=int(np.prod(ng)), mask=mask,
                  ub=ub, spng=spng, re=re, endtime=endtime, cfl=cfl, lxd=3 * n // 2, has_outflow=has_out, meta=meta)
