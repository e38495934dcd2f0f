"""Time-stepper parity through the C-ABI: nsk_matvec vs the oracle on identical
inputs (a committed reference eigenmode as the initial perturbation)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mode(o, modes, key):
    J = o.J12
    u = modes[key + "_u"].astype(np.float64)
    p = modes[key + "_p"].astype(np.float64)
    return u[0], u[1], J @ p @ J.T


def relL2(o, a, b):
    w = o.bm1
    num = sum(np.sum(w * (x - y) ** 2) for x, y in zip(a[:2], b[:2]))
    den = sum(np.sum(w * y ** 2) for y in b[:2])
    return np.sqrt(num / den)


@pytest.mark.parametrize("mode,nsteps", [(0, 5), (1, 5)])
def test_steps_vs_oracle(hip6, oracle6, modes, mode, nsteps):
    """A few linearised steps with tight solver tolerances: GPU == oracle to ~1e-10
    (velocity, mass-weighted L2) and pressure to 1e-7."""
    o = oracle6
    q = _mode(o, modes, "dRe")
    hip6.set_tolerances(1e-13, 1e-13, 0)
    hip6.set_nsteps(nsteps)
    vq, vf = hip6.alloc(2)
    hip6.upload(vq, *q)
    hip6.matvec(vf, vq, mode)
    f = hip6.download(vf)
    ref = o.matvec(q, adjoint=bool(mode), nsteps=nsteps)
    hip6.set_nsteps(100)
    print("stats", hip6.stats())
    assert relL2(o, f, ref) < 1e-9
    assert np.abs(f[2] - ref[2]).max() / np.abs(ref[2]).max() < 1e-5
    hip6.free([vq, vf])


@pytest.fixture(scope="module")
def adjoint_setup():
    """The reference's adjoint case: 'O' -> 'v' (1cyl.usr:126-132) => all-Dirichlet/periodic
    velocity, singular pressure operator (`ortho`)."""
    import os
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from tests.conftest import GOLDEN, make_oracle
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, adjoint=True)
    assert not case.has_outflow
    o = make_oracle(case)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-13, tol_pres=1e-6, tol_relative=1,
                   max_helm_iter=120, max_pres_iter=48)
    yield case, o, h
    h.close()


def test_adjoint_case_steps_vs_oracle(adjoint_setup, modes):
    case, o, h = adjoint_setup
    u = modes["dRe_u"].astype(np.float64)              # any smooth C0 field that satisfies the BCs
    q = (u[0] * case.mask, u[1] * case.mask, o.J12 @ modes["dRe_p"].astype(np.float64) @ o.J12.T)
    h.set_nsteps(6)
    vq, vf = h.alloc(2)
    h.upload(vq, *q)
    h.matvec(vf, vq, 1)
    f = h.download(vf)
    ref = o.matvec(q, adjoint=True, nsteps=6)
    assert relL2(o, f, ref) < 1e-9
    dp = (f[2] - f[2].mean()) - (ref[2] - ref[2].mean())       # pressure level is arbitrary
    assert np.abs(dp).max() / np.abs(ref[2] - ref[2].mean()).max() < 1e-5


def test_adjoint_eigen_relation_lx1_8(modes, spectre):
    """KAT for the adjoint map at lx1=8 (the only N=7 pin): Rayleigh quotient of the GPU matvec on
    the reference's aRe/aIm (file index 2 = conjugate partner) equals Spectre_Ha.dat:2."""
    import os
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    from tests.conftest import GOLDEN
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8, adjoint=True)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-2, tol_relative=1,
                   max_helm_iter=120, max_pres_iter=48)
    assert h.nsteps == 183
    J = interp_matrix(gauss_lobatto_legendre(8)[0], gauss_legendre(6)[0])
    vr, vi, fr, fi = h.alloc(4)
    for v, k in ((vr, "aRe"), (vi, "aIm")):
        u = modes[k + "_u"].astype(np.float64)
        h.upload(v, u[0], u[1], J @ modes[k + "_p"].astype(np.float64) @ J.T)
    h.matvec(fr, vr, 1); h.matvec(fi, vi, 1)
    ray = (h.dot(vr, fr) + h.dot(vi, fi)) + 1j * (h.dot(vr, fi) - h.dot(vi, fr))
    mu = complex(spectre["Ha"][1, 0], spectre["Ha"][1, 1])
    print("adjoint rayleigh", ray, "reference", mu)
    assert abs(ray - mu) < 5e-7
    h.close()


def test_composed_maps(hip6, oracle6, modes):
    """transient_growth_map = adjoint(forward(q)) (core/matvec.f:343-346) and
    newton_linearized_map = forward(q) - q (core/matvec.f:398-401)."""
    o = oracle6
    q = _mode(o, modes, "dRe")
    hip6.set_tolerances(1e-13, 1e-6, 1)
    hip6.set_nsteps(3)
    vq, vf = hip6.alloc(2)
    hip6.upload(vq, *q)
    hip6.matvec(vf, vq, 2)
    f = hip6.download(vf)
    ref = o.matvec(o.matvec(q, adjoint=False, nsteps=3), adjoint=True, nsteps=3)
    assert relL2(o, f, ref) < 1e-9
    hip6.matvec(vf, vq, 3)
    f = hip6.download(vf)
    fw = o.matvec(q, adjoint=False, nsteps=3)
    ref = tuple(a - b for a, b in zip(fw, q))
    num = np.sqrt(sum(np.sum(o.bm1 * (x - y) ** 2) for x, y in zip(f[:2], ref[:2])))
    den = np.sqrt(sum(np.sum(o.bm1 * y ** 2) for y in fw[:2]))
    assert num / den < 1e-9
    # ts_force_sensitivity_map = (I - exp(L^+ T)) q  (core/matvec.f:357-374, uparam(1) = 4)
    hip6.matvec(vf, vq, 4)
    f = hip6.download(vf)
    ad = o.matvec(q, adjoint=True, nsteps=3)
    ref = tuple(b - a for a, b in zip(ad, q))
    num = np.sqrt(sum(np.sum(o.bm1 * (x - y) ** 2) for x, y in zip(f[:2], ref[:2])))
    den = np.sqrt(sum(np.sum(o.bm1 * y ** 2) for y in ad[:2]))
    assert num / den < 1e-9
    hip6.set_nsteps(100)
    hip6.set_tolerances(1e-13, 1e-13, 0)           # the session fixture's settings (tests that follow rely on them)
    hip6.free([vq, vf])
