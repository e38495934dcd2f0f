"""Backward-facing step at Re = 500 (examples/back_fstep/transient_growth, the geometry of BASELINE config 4): a
gmsh-made ``.re2`` (format v003, boundary ids), an all-Dirichlet perturbation problem with sponges, and the
reference's committed transient-growth results for T = 1 -- the optimal initial perturbation ``pRebfs0.f00001`` and
its response ``orebfs0.f00001 = M pRe`` (core/eigensolvers.f:645-652).  Pins the direct map, the direct-adjoint
composition (core/matvec.f:332-349) and, through z-extrusion, the hexahedral path on this geometry."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

KW = dict(tol_helm=1e-11, tol_pres=1e-5, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=192)     # (closed domain: some projected solves need a second GMRES cycle)
KW3 = dict(KW, tol_pres=1e-3)      # hexahedral FDM-Schwarz, closed domain (no projection space): 26 iterations per step at 1e-3


@pytest.fixture(scope="module")
def bfs():
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    case = mesh.load_case_npz(os.path.join(GOLDEN, "backstep_case.npz"), 6, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0,
                              spng_str=2.0)              # bfs.par: userParam08-10, viscosity -500, endTime 1
    tg = np.load(os.path.join(GOLDEN, "backstep_tg.npz"))
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], **KW)
    yield case, tg, J, h
    h.close()


def _wnorm2(w, a):
    return float(np.sum(w * (a[0] ** 2 + a[1] ** 2)))


def test_backstep_setup_pins(bfs):
    case, tg, J, h = bfs
    assert not case.has_outflow                       # usrdat2: ids 4, 2 -> 'v', 3 -> 'W'
    assert h.nsteps == 172 and int(tg["pRe_istep"]) == 173          # field header: istep = nsteps + 1
    from nekstab_amd.quadrature import gauss_lobatto_legendre
    # unit norm of the committed optimal perturbation under the sponge-masked inner product
    v = h.alloc(1)[0]
    h.upload(v, tg["pRe_u"][0].astype(float), tg["pRe_u"][1].astype(float), np.zeros(h.npres))
    assert abs(h.norm(v) ** 2 - 1.0) < 1e-6
    assert np.abs(tg["pIm_u"]).max() == 0.0           # M^T M is symmetric: real eigenvectors


def test_direct_map_reproduces_committed_optimal_response(bfs):
    case, tg, J, h = bfs
    q, f = h.alloc(2)
    pu = tg["pRe_u"].astype(float)
    h.upload(q, pu[0], pu[1], J @ tg["pRe_p"].astype(float) @ J.T)
    h.matvec(f, q, 0)
    out = h.download(f)
    ore = tg["ore_u"].astype(float)
    from oracle.linns import LinNS2D
    o = LinNS2D(x=case.x, y=case.y, gid=case.gid, nglob=case.nglob, mask=case.mask, ub=case.ub, spng=case.spng, re=case.re,
                endtime=case.endtime, has_outflow=False, build_solvers=False)
    err = np.sqrt(_wnorm2(o.bm1, [out[0] - ore[0], out[1] - ore[1]]) / _wnorm2(o.bm1, ore))
    assert err < 2e-5, err                            # fp32 files, eigen_tol 1e-6
    gain = h.norm(f) ** 2
    assert abs(gain - _wnorm2(o.bm1s(), ore)) < 2e-5 * gain
    assert abs(gain - 3.2370) < 1e-3                  # optimal energy gain G(T = 1)


def test_transient_growth_eigen_relation(bfs):
    from nekstab_amd.capi import NSK_DIRECT_ADJOINT
    case, tg, J, h = bfs
    q, g = h.alloc(2)
    pu = tg["pRe_u"].astype(float)
    h.upload(q, pu[0], pu[1], J @ tg["pRe_p"].astype(float) @ J.T)
    h.matvec(g, q, NSK_DIRECT_ADJOINT)                # M^T M pRe = G pRe
    lam = h.dot(q, g) / h.dot(q, q)
    assert abs(lam - 3.2370) < 2e-3, lam
    h.axpy(g, -lam, q)
    assert h.norm(g) < 2e-3 * lam, h.norm(g)          # residual of the committed eigenvector (fp32 storage)


def test_hexahedral_path_on_the_extruded_step(bfs):
    """config 4's geometry: the step mesh extruded in z, periodic; z-invariant input => the reference's response."""
    from nekstab_amd import mesh3d
    from nekstab_amd.capi import NekStabHip
    case, tg, J, h = bfs
    c3 = mesh3d.extrude_case(case, 2, 1.0, periodic=True)
    h3 = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], **KW3)
    try:
        assert h3.nsteps == 172
        pu = tg["pRe_u"].astype(float)
        q, f = h3.alloc(2)
        h3.upload3(q, mesh3d.extrude_field(pu[0], 2), mesh3d.extrude_field(pu[1], 2), np.zeros(c3.x.shape),
                   mesh3d.extrude_pressure(J @ tg["pRe_p"].astype(float) @ J.T, 2))
        h3.matvec(f, q, 0)
        out = h3.download3(f)
        ore = tg["ore_u"].astype(float)
        sc = np.abs(ore).max()
        for layer in range(2):
            e = slice(layer * case.nel, (layer + 1) * case.nel)
            for k in (0, 3, 5):
                assert np.abs(out[0][e, k] - ore[0]).max() < 2e-6 * sc
                assert np.abs(out[1][e, k] - ore[1]).max() < 2e-6 * sc
        assert np.abs(out[2]).max() < 1e-7 * sc
        assert abs(h3.norm(f) ** 2 - 3.2370) < 1e-3  # span 1.0: same energy gain
        # config 4 is an *adjoint* run: direct-adjoint composition on hexahedra, M^T M pRe = G pRe
        from nekstab_amd.capi import NSK_DIRECT_ADJOINT
        g = h3.alloc(1)[0]
        h3.matvec(g, q, NSK_DIRECT_ADJOINT)
        lam = h3.dot(q, g) / h3.dot(q, q)
        assert abs(lam - 3.2370) < 2e-3, lam
        h3.axpy(g, -lam, q)
        assert h3.norm(g) < 3e-3 * lam
    finally:
        h3.close()
