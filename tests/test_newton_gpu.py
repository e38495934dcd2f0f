"""Nonlinear time-stepper map and the Newton-Krylov fixed-point solver on the device,
pinned on the reference's data: its converged Re=50 base flow is a fixed point of the map to the Newton
tolerance of the run that produced it (residualTol 1e-11, examples/cylinder/baseflow/newton/1cyl.par), and
Newton started from its Re=40 field converges to that base flow."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN, make_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nosponge():
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, spng_str=0.0)   # the Newton example has no sponge
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-3, tol_relative=1,
                   nproj=8, max_helm_iter=150, max_pres_iter=96)     # 1e-8 solves sit at 44-46 iterations: leave room for a restart
    yield case, h
    h.close()


def _bf_state(case):
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    return case.ub[0], case.ub[1], J @ case.meta["bf_p"] @ J.T


def test_nonlinear_steps_vs_oracle(nosponge):
    case, h = nosponge
    o = make_oracle(case)
    q = _bf_state(case)
    q = (q[0] * (1 + 0.05 * np.sin(case.y)), q[1] + 0.02 * np.cos(case.x) * case.mask, q[2])   # away from the fixed point
    vq, vf = h.alloc(2)
    h.upload(vq, *q)
    h.set_baseflow(vq)
    o.set_baseflow(np.stack(q[:2]))
    assert h.nsteps == o.nsteps and abs(h.dt - o.dt) < 1e-15
    h.set_tolerances(1e-13, 1e-8, 1)            # the perturbed start is far from solenoidal: converge the projection
    h.set_nsteps(5)
    h.nonlinear_map(vf, vq)
    f = h.download(vf)
    ref = o.nonlinear_map(q, nsteps=5)
    num = np.sqrt(sum(np.sum(o.bm1 * (a - b) ** 2) for a, b in zip(f[:2], ref[:2])))
    den = np.sqrt(sum(np.sum(o.bm1 * b ** 2) for b in ref[:2]))
    assert num / den < 1e-9
    # the linearised (newton) map about the same state is the derivative of the nonlinear one
    eps = 1e-5
    dq = (0.3 * np.sin(case.x) * case.mask, 0.2 * np.cos(case.y) * case.mask, np.zeros_like(q[2]))
    vp, vfp, vl, vd = h.alloc(4)
    h.upload(vp, q[0] + eps * dq[0], q[1] + eps * dq[1], q[2])
    h.nonlinear_map(vfp, vp)
    h.upload(vd, *dq)
    h.matvec(vl, vd, 0)
    fp, lin = h.download(vfp), h.download(vl)
    fd = [(a - b) / eps for a, b in zip(fp[:2], f[:2])]
    err = np.sqrt(sum(np.sum(o.bm1 * (a - b) ** 2) for a, b in zip(fd, lin[:2]))) / np.sqrt(sum(np.sum(o.bm1 * b ** 2) for b in lin[:2]))
    print("finite-difference vs linearised map:", err)
    assert err < 1e-3
    h.set_tolerances(1e-12, 1e-3, 1)
    h.free([vq, vf, vp, vfp, vl, vd])


def test_reference_baseflow_is_a_fixed_point(nosponge):
    case, h = nosponge
    q = _bf_state(case)
    vq, vf = h.alloc(2)
    h.upload(vq, *q)
    h.set_baseflow(vq)
    assert h.nsteps == 100
    h.nonlinear_map(vf, vq, subtract_q=True)
    res = h.norm(vf) ** 2
    print("|Phi_T(BF) - BF|^2 =", res, " |BF|^2 =", h.norm(vq) ** 2)
    assert res < 5e-11                                   # reference Newton tolerance: 1e-11
    h.free([vq, vf])


def test_newton_from_re40_reaches_reference_baseflow(nosponge):
    from nekstab_amd import mesh, newton
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    case, h = nosponge
    z = np.load(os.path.join(GOLDEN, "cylinder_bf_re40.npz"))
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    q = h.alloc(1)[0]
    h.upload(q, z["u"][0].astype(np.float64), z["u"][1].astype(np.float64), J @ z["p"].astype(np.float64) @ J.T)
    its, hist = newton.newton_krylov(h, q, k_dim=100, tol=1e-11, maxiter_newton=12,
                                     log=lambda kind, i, r: print(kind, i, "%.3e" % r) if kind != "arnoldi" else None)
    assert hist[-1] < 1e-11 and its <= 10
    got = h.download(q)
    o_w = None
    w = h  # noqa
    ref = case.ub
    num = np.sqrt(np.sum((got[0] - ref[0]) ** 2 + (got[1] - ref[1]) ** 2))
    den = np.sqrt(np.sum(ref[0] ** 2 + ref[1] ** 2))
    print("Newton iterations", its, "distance to the committed base flow", num / den)
    assert num / den < 1e-6
    h.free([q])


def test_newton_for_periodic_orbit_refines_the_reference_upo():
    """Newton-GMRES for unstable periodic orbits (uparam(1) = 2.1; core/newton_krylov.f, newton_linearized_map's bordered
    branch core/matvec.f:402-418): started from the reference's converged vortex-shedding orbit (the base flow of its Floquet
    example, period 7.921338 in its header) with the period of examples/cylinder/baseflow/newton_upo's `.par` (7.9), the
    iteration closes the orbit (|Phi_T(q) - q|^2 below 1e-9) and returns the reference's period."""
    import os
    from nekstab_amd import mesh, newton
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cylinder_upo.npz"))
    T_ref = float(z["period"])
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, endtime=7.9)
    case.ub[:] = z["u"]
    case.spng[:] = 0.0                                         # newton_upo/1cyl.par sets no sponge
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-3, tol_relative=1,
                   nproj=8, max_helm_iter=150, max_pres_iter=48)
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    q = h.alloc(1)[0]
    h.upload(q, z["u"][0], z["u"][1], J @ z["p"] @ J.T)
    log = []
    T, its, hist = newton.newton_krylov_upo(h, q, 7.9, k_dim=60, tol=1e-9, maxiter_newton=6,
                                            log=lambda *a: log.append(a) if a[0] == "newton" else None)
    print("Newton UPO: period %.6f (reference %.6f) after %d iterations; residual^2 / period history:" % (T, T_ref, its), hist)
    assert hist[0][0] > 1e-6                                   # T = 7.9 does not close the orbit
    assert hist[-1][0] < 1e-9 and its <= 6
    assert abs(T - T_ref) < 2e-5 * T_ref                       # measured: 7.921337 against the header's 7.921338
    h.close()
