"""Kernel parity at the other polynomial orders the reference's SIZE lists (lx1 = 8, 10, 12;
lxd = 12, 15, 18) on the cylinder mesh -- oracle operators only, no solves."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("lx1", [8, 10, 12])
def test_operators(lx1):
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from tests.conftest import GOLDEN, make_oracle
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), lx1)
    o = make_oracle(case, build_solvers=False)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], schwarz_layers=1)
    assert (h.nsteps, abs(h.dt - o.dt) < 1e-15) == (o.nsteps, True)
    rng = np.random.default_rng(lx1)
    u = np.sin(0.7 * case.x) * np.cos(0.5 * case.y) + 0.1 * rng.standard_normal(case.x.shape)
    v = np.cos(0.3 * case.x) * np.sin(0.9 * case.y) + 0.1 * rng.standard_normal(case.x.shape)
    p = rng.standard_normal((case.nel, lx1 - 2, lx1 - 2))
    assert rel(h.t_axhelm(u, 0.02, 300.0), o.axhelm(u, 0.02, 300.0)) < 1e-12
    assert rel(h.t_dssum(u), o.dssum(u)) < 1e-13
    assert rel(h.t_opdiv(u, v), o.opdiv(u, v)) < 1e-12
    gx, gy = h.t_opgradt(p); ox, oy = o.opgradt(p)
    assert rel(gx, ox) < 1e-12 and rel(gy, oy) < 1e-12
    U, V = o.ub
    bx = -o.spng * u * o.bm1 - (o.convect(u, v, U) + o.convect(U, V, u))
    by = -o.spng * v * o.bm1 - (o.convect(u, v, V) + o.convect(U, V, v))
    cx, cy = h.t_convect(u, v, False)
    assert rel(cx, bx) < 1e-12 and rel(cy, by) < 1e-12
    wx, wy = o.opgradt(p)
    fac = o.binvm1 * o.mask
    ref = o.opdiv(fac * o.dssum(wx * o.mask), fac * o.dssum(wy * o.mask))
    assert rel(h.t_eapply(p), ref) < 1e-12
    # one full time step through the solvers is exercised at lx1=6/8 in test_matvec_gpu.py
    h.close()


@pytest.mark.parametrize("lx1", [10, 12])
def test_full_steps_other_orders(lx1, modes):
    """Whole time steps (all solver kernels) at lx1 = 10 and 12: the result is finite, discretely
    divergence free to the solver tolerance, and its energy agrees with the lx1=6 oracle-checked
    result to the resolution difference (regression for the lx1=12 staging bug)."""
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    from tests.conftest import GOLDEN
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), lx1)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-3, tol_relative=1,
                   nproj=8, max_helm_iter=150, max_pres_iter=48)
    u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1)
    p1 = mesh.interp_field_2d(modes["dRe_p"].astype(np.float64), lx1)
    J = interp_matrix(gauss_lobatto_legendre(lx1)[0], gauss_legendre(lx1 - 2)[0])
    vq, vf = h.alloc(2)
    h.upload(vq, u[0], u[1], J @ p1 @ J.T)
    e0 = h.dot(vq, vq)
    h.set_nsteps(6)
    h.matvec(vf, vq, 0)
    f = h.download(vf)
    assert all(np.all(np.isfinite(a)) for a in f)
    div = h.t_opdiv(f[0], f[1])
    div0 = h.t_opdiv(u[0] + 0.01 * case.x * case.mask, u[1])          # scale of a non-solenoidal field
    assert np.abs(div).max() < 1e-6 * np.abs(div0).max()
    e1 = h.dot(vf, vf)
    # the reference mode decays/rotates with |mu|^(2*6*dt): energy ratio close to that of the eigenvalue
    mu_abs = abs(0.7387113 + 0.6972442j)
    assert abs(e1 / e0 - mu_abs ** (2 * 6 * h.dt)) < 2e-3
    h.close()


def test_refined_mesh_config3_paths(modes):
    """BASELINE config 3's mesh (every element split 2x2, E = 7984): exercises the streaming coarse solve (8k vertices)
    and, with more than 1024 workgroups, the summed-once totals path of the quadrilateral kernels.  Operators against
    the oracle on the refined mesh; a few whole time steps on the reference's eigenmode."""
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from tests.conftest import GOLDEN, make_oracle
    c0 = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    case = mesh.refine_case_2x2(c0)
    assert case.nel == 4 * c0.nel
    o = make_oracle(case, build_solvers=False)
    kw = dict(tol_helm=1e-11, tol_pres=1e-4, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=48)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], **kw)
    try:
        rng = np.random.default_rng(0)
        u = np.sin(0.7 * case.x) * np.cos(0.5 * case.y) + 0.1 * rng.standard_normal(case.x.shape)
        p = rng.standard_normal((case.nel, 4, 4))
        assert rel(h.t_axhelm(u, 0.02, 300.0), o.axhelm(u, 0.02, 300.0)) < 1e-12
        assert rel(h.t_dssum(u), o.dssum(u)) < 1e-13
        wx, wy = o.opgradt(p)
        fac = o.binvm1 * o.mask
        assert rel(h.t_eapply(p), o.opdiv(fac * o.dssum(wx * o.mask), fac * o.dssum(wy * o.mask))) < 1e-12
        # a few whole time steps (all solver kernels of these paths) on the reference's eigenmode: converged, and the
        # energy follows the eigenvalue (|mu|^(2 t / T)), as in test_full_steps_other_orders
        m = modes["dRe_u"].astype(np.float64)
        cm = mesh.refine_case_2x2(mesh.Case(**{**c0.__dict__, "ub": m}))
        nst = 6
        b0, b1 = h.alloc(2)
        h.upload(b0, cm.ub[0] * case.mask, cm.ub[1] * case.mask, np.zeros(h.npres))
        e0 = h.dot(b0, b0)
        h.set_nsteps(nst); h.matvec(b1, b0, 0)
        e1 = h.dot(b1, b1)
        assert h.stats()["unconverged"] == 0
        mu_abs = abs(0.7387113 + 0.6972442j)
        assert abs(e1 / e0 - mu_abs ** (2 * nst * h.dt)) < 2e-3
        r1 = h.download(b1)
        div = h.t_opdiv(r1[0], r1[1])
        div0 = h.t_opdiv(cm.ub[0] + 0.01 * case.x * case.mask, cm.ub[1])
        assert np.abs(div).max() < 1e-5 * np.abs(div0).max()
    finally:
        h.close()
