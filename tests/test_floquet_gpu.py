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
