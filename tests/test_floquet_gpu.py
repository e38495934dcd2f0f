"""Floquet analysis of the periodic vortex-shedding orbit (examples/cylinder/stability/direct_Floquet,
uparam(1)=3.11): base flow integrated over one period on the device and stored per step; the
Floquet multipliers of the k_dim=100 Arnoldi are compared with the reference's Spectre_Hd.dat."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_floquet_multipliers():
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    z = np.load(os.path.join(GOLDEN, "cylinder_upo.npz"))
    T = float(z["period"])
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, endtime=T)
    case.ub[:] = z["u"]                                             # the UPO snapshot is the initial base flow
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-2, tol_relative=1,
                   nproj=8, max_helm_iter=150, max_pres_iter=48)
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    q0, qe = h.alloc(2)
    h.upload(q0, z["u"][0], z["u"][1], J @ z["p"] @ J.T)
    h.set_orbit(q0, spng_str=1.7, end=qe)
    assert h.nsteps == 795                                          # field header: istep 796
    h.axpy(qe, -1.0, q0)
    per = h.norm(qe) / h.norm(q0)
    print("periodicity |Phi_T(q0) - q0| / |q0| =", per)
    assert per < 5e-3                                               # the reference orbit closes to ~1e-3 (multiplier 1.000846)
    qx, qy = seed.add_noise(case)
    v0, v1 = h.alloc(2)
    h.upload(v0, qx, qy, np.zeros(h.npres))
    h.scal(v0, 1.0 / h.norm(v0))
    h.matvec(v1, v0, 0)
    res = krylov.krylov_schur(h, v1, 40, schur_tgt=0)
    ref = z["spectre_Hd"]
    # leading multipliers of the reference table (its rows 2-5 are break-down artefacts: exactly 1.0 with
    # residual 1e-17 after the Krylov space became invariant)
    want = [complex(1.000846, 0.0), complex(0.8117152, 0.0), complex(0.2760130, 0.04737052)]
    for w in want:
        j = np.argmin(np.abs(res.vals - w))
        print("reference", w, "ours", res.vals[j], "residual", res.residual[j])
    for w, tol in zip(want, (2e-5, 2e-5, 1e-3)):
        j = np.argmin(np.abs(res.vals - w))
        assert abs(res.vals[j] - w) < tol
    h.close()


def test_adjoint_floquet_lx1_8():
    """uparam(1)=3.21 at lx1=8 (examples/cylinder/stability/adjoint_Floquet): adjoint operator over the
    forward-integrated periodic base flow, outflow boundary unchanged (1cyl.usr:126 only switches for 3.2).
    Leading pair of the reference table: 0.9662098 -+ 0.0073784 i."""
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    z = np.load(os.path.join(GOLDEN, "cylinder_upo.npz"))
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8, endtime=float(z["period"]))
    u = mesh.interp_field_2d(z["u"], 8)
    p1 = mesh.interp_field_2d(z["p"], 8)
    case.ub[:] = u
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-2, tol_relative=1,
                   nproj=8, max_helm_iter=150, max_pres_iter=48)
    J = interp_matrix(gauss_lobatto_legendre(8)[0], gauss_legendre(6)[0])
    q0 = h.alloc(1)[0]
    h.upload(q0, u[0], u[1], J @ p1 @ J.T)
    h.set_orbit(q0, spng_str=1.7)
    qx, qy = seed.add_noise(case)
    v0, v1 = h.alloc(2)
    h.upload(v0, qx, qy, np.zeros(h.npres))
    h.scal(v0, 1.0 / h.norm(v0))
    h.matvec(v1, v0, 1)
    res = krylov.krylov_schur(h, v1, 28, mode=1, schur_tgt=0)
    ref = complex(z["spectre_Ha"][5, 0], abs(z["spectre_Ha"][5, 1]))
    j = np.argmin(np.abs(res.vals - ref))
    print("reference", ref, "ours", res.vals[j], "residual", res.residual[j])
    assert abs(res.vals[j] - ref) < 2e-5
    h.close()


@pytest.mark.parametrize("lx1", [6, 8])
def test_time_periodic_base_flow_on_hexahedra_equals_the_quadrilateral_path(lx1):
    """Time-periodic base flows on hexahedra (core/matvec.f:191-236; missing until round 4): `nsk_set_orbit` on a hexahedral context
    integrates the full equations from the given state and stores the twelve dealiasing-mesh base-flow constants of every time
    step; direct and adjoint maps then read slot `istep`.  Pinned through the 2-D path (itself pinned on the reference's Floquet
    multipliers above): on the cylinder mesh extruded over two periodic layers, with the z-invariant vortex-shedding state as the
    initial base flow and a z-invariant perturbation, the hexahedral orbit and maps reproduce the quadrilateral ones plane by
    plane -- over a 40-step stretch of the orbit, which exercises every slot.  lx1 = 6 runs the LDS convection kernels, lx1 = 8 the
    matrix-core one (the shedding state interpolated to the finer points: both paths start from the same field)."""
    from nekstab_amd import mesh, mesh3d, seed
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    z = np.load(os.path.join(GOLDEN, "cylinder_upo.npz"))
    kw = dict(tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, nproj=0, max_helm_iter=400, max_pres_iter=192)
    c2 = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), lx1, endtime=0.4)
    Ju = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_lobatto_legendre(lx1)[0])
    uo = np.stack([Ju @ z["u"][k] @ Ju.T for k in range(2)]) if lx1 != 6 else z["u"]
    c2.ub[:] = uo
    nz = 2
    c3 = mesh3d.extrude_case(c2, nz, 0.5, periodic=True)
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(lx1 - 2)[0])
    p2 = J @ z["p"] @ J.T
    h2 = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], **kw)
    h3 = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], **kw)
    try:
        a2, e2 = h2.alloc(2)
        h2.upload(a2, uo[0], uo[1], p2)
        h2.set_orbit(a2, spng_str=1.7, end=e2)
        a3, e3 = h3.alloc(2)
        h3.upload3(a3, mesh3d.extrude_field(uo[0], nz), mesh3d.extrude_field(uo[1], nz), np.zeros(c3.x.shape),
                   mesh3d.extrude_pressure(p2, nz))
        h3.set_orbit(a3, spng_str=1.7, end=e3)
        assert h3.nsteps == h2.nsteps and abs(h3.dt - h2.dt) < 1e-15 and h2.nsteps >= 30
        u2 = h2.download(e2); u3 = h3.download3(e3)
        sc = max(np.abs(u2[0]).max(), np.abs(u2[1]).max())
        for k in range(2):
            assert np.abs(u3[k] - mesh3d.extrude_field(u2[k], nz)).max() < 1e-8 * sc          # the orbit itself
        assert np.abs(u3[2]).max() < 1e-8 * sc
        qx, qy = seed.add_noise(c2)
        for mode in (0, 1):
            v2, f2 = h2.alloc(2); v3, f3 = h3.alloc(2)
            h2.upload(v2, qx, qy, np.zeros(h2.npres))
            h3.upload3(v3, mesh3d.extrude_field(qx, nz), mesh3d.extrude_field(qy, nz), np.zeros(c3.x.shape), np.zeros(h3.npres))
            h2.matvec(f2, v2, mode); h3.matvec(f3, v3, mode)
            r2 = h2.download(f2); r3 = h3.download3(f3)
            sc = max(np.abs(r2[0]).max(), np.abs(r2[1]).max())
            err = max(np.abs(r3[k] - mesh3d.extrude_field(r2[k], nz)).max() for k in range(2)) / sc
            print("mode", mode, "hexahedral vs quadrilateral map over the stored orbit:", err)
            assert err < 1e-6 and np.abs(r3[2]).max() < 1e-7 * sc
        # a map longer than the stored orbit is refused, as in 2-D
        h3.set_nsteps(h3.nsteps + 5)
        with pytest.raises(Exception):
            h3.matvec(f3, v3, 0)
    finally:
        h2.close(); h3.close()
