"""Pins of the CPU oracle against the reference's committed data:
field-header step counts, mass-matrix sum, mode normalisation, the eigen-relation of the
reference's own eigenmode with the reference's own eigenvalue."""
import numpy as np
import pytest

from tests.conftest import make_oracle


def test_nsteps_and_dt_lx1_6(oracle6_nosolve, modes):
    o = oracle6_nosolve
    assert o.nsteps == 100 and abs(o.dt - 0.01) < 1e-15
    assert int(modes["dRe_istep"]) == o.nsteps + 1           # header holds the DO-loop exit value


def test_nsteps_lx1_8(modes):
    from nekstab_amd import mesh
    import os
    from tests.conftest import GOLDEN
    c8 = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8)
    o = make_oracle(c8, build_solvers=False)
    assert o.nsteps == 183
    assert int(modes["aRe_istep"]) == o.nsteps + 1
    # mode normalisation at lx1=8 (adjoint files): |Re|^2 + |Im|^2 = 1 under bm1s
    a = (modes["aRe_u"][0].astype(float), modes["aRe_u"][1].astype(float))
    b = (modes["aIm_u"][0].astype(float), modes["aIm_u"][1].astype(float))
    assert abs(o.inner(a, a) + o.inner(b, b) - 1.0) < 1e-6


def test_mass_matrix_and_mode_norm(oracle6_nosolve, modes):
    o = oracle6_nosolve
    assert abs(o.volvm1 - (66.0 * 32.0 - np.pi / 4.0)) < 1e-6   # box minus the unit-diameter cylinder
    a = (modes["dRe_u"][0].astype(float), modes["dRe_u"][1].astype(float))
    b = (modes["dIm_u"][0].astype(float), modes["dIm_u"][1].astype(float))
    assert abs(o.inner(a, a) + o.inner(b, b) - 1.0) < 1e-6      # core/eigensolvers.f:619-622
    w = o.bm1                                                    # without the sponge mask the norm is ~1.018
    full = sum(np.sum(x * w * x) for x in a + b)
    assert full > 1.01


@pytest.mark.slow
def test_eigen_relation_direct(oracle6, modes, spectre):
    """M (dRe + i dIm) = mu (dRe + i dIm): Rayleigh quotient of the oracle's matvec on the
    reference's eigenmode equals the reference's eigenvalue to its 7 printed digits."""
    o = oracle6
    J = o.J12
    mu = complex(spectre["Hd"][0, 0], spectre["Hd"][0, 1])
    qr = (modes["dRe_u"][0].astype(float), modes["dRe_u"][1].astype(float), J @ modes["dRe_p"].astype(float) @ J.T)
    qi = (modes["dIm_u"][0].astype(float), modes["dIm_u"][1].astype(float), J @ modes["dIm_p"].astype(float) @ J.T)
    fr, fi = o.matvec(qr), o.matvec(qi)
    ray = (o.inner(qr, fr) + o.inner(qi, fi)) + 1j * (o.inner(qr, fi) - o.inner(qi, fr))
    assert abs(ray - mu) < 2e-7
    er = [fr[k] - (mu.real * qr[k] - mu.imag * qi[k]) for k in range(2)]
    ei = [fi[k] - (mu.imag * qr[k] + mu.real * qi[k]) for k in range(2)]
    assert np.sqrt(o.inner(er, er) + o.inner(ei, ei)) < 1e-5      # fp32 mode storage + eigen_tol 1e-6


def test_spectre_tables_consistent(spectre):
    """lambda = log(mu)/T links Spectre_H and Spectre_NS (core/eigensolvers.f:593-595)."""
    from nekstab_amd.krylov import log_transform
    for op in ("d", "a"):
        H, NS = spectre["H" + op], spectre["NS" + op]
        lam = log_transform(H[:, 0] + 1j * H[:, 1], 1.0)
        ok = np.abs(H[:, 0] + 1j * H[:, 1]) > 1e-3
        assert np.abs(lam.real[ok] - NS[ok, 0]).max() < 2e-5
        assert np.abs(np.abs(lam.imag[ok]) - np.abs(NS[ok, 1])).max() < 2e-5


@pytest.mark.slow
def test_reference_baseflow_is_fixed_point_of_nonlinear_map():
    """nonlinear_forward_map pin (core/newton_krylov.f:336-378): the reference's converged Re=50 base flow
    (written by its Newton run at residualTol 1e-11) satisfies |Phi_T(BF) - BF|^2 ~ 1e-11 under the oracle's
    nonlinear step -- an O(1) error in the nonlinear discretisation would give O(1e-3)."""
    import os
    from nekstab_amd import mesh
    from tests.conftest import GOLDEN
    c = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, spng_str=0.0)
    o = make_oracle(c)
    J = o.J12
    q = (c.ub[0], c.ub[1], J @ c.meta["bf_p"] @ J.T)
    f = o.nonlinear_map(q)
    d = [f[0] - q[0], f[1] - q[1]]
    assert o.inner(d, d) < 5e-11


def test_oracle_pins_on_the_other_reference_geometries():
    """dt rule, geometry and the sponge-masked inner product of the oracle on the reference's two other 2-D examples:
    backward-facing step (transient growth, field headers istep = 173, unit-norm optimal perturbation, gain = |ore|^2)
    and lid-driven cavity (base-flow header istep = 697)."""
    import os
    from nekstab_amd import mesh, nekio
    from oracle.linns import LinNS2D
    here = os.path.dirname(os.path.abspath(__file__))
    c = mesh.load_case_npz(os.path.join(here, "golden", "backstep_case.npz"), 6, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
    o = LinNS2D(x=c.x, y=c.y, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re, endtime=c.endtime,
                has_outflow=c.has_outflow, build_solvers=False)
    tg = np.load(os.path.join(here, "golden", "backstep_tg.npz"))
    assert o.nsteps == 172 and int(tg["pRe_istep"]) == 173 and int(tg["ore_istep"]) == 173
    w = o.bm1s()
    n2 = lambda a: float(np.sum(w * (a[0].astype(float) ** 2 + a[1].astype(float) ** 2)))
    assert abs(n2(tg["pRe_u"]) - 1.0) < 1e-7                     # core/eigensolvers.f:619-627
    assert abs(n2(tg["ore_u"]) - 3.2370) < 1e-3                  # optimal energy gain G(T=1) of the committed response
    assert not c.has_outflow and c.spng.max() == 1.0
    z = np.load(os.path.join(here, "golden", "cavity_case.npz"))
    bcs = [(int(a), int(b), np.zeros(5), str(cd)) for (a, b), cd in zip(z["bc_ef"], z["bc_code"])]
    m = nekio.Re2Mesh(2, z["xc"].shape[0], z["xc"], z["yc"], None, [], bcs)
    c2 = mesh.build_case_2d(m, z["vlex"].astype(np.int64), z["bf_u"].astype(np.float64), 6, re=3600.0, endtime=1.0, spng_str=0.0)
    o2 = LinNS2D(x=c2.x, y=c2.y, gid=c2.gid, nglob=c2.nglob, mask=c2.mask, ub=c2.ub, spng=c2.spng, re=c2.re, endtime=c2.endtime,
                 has_outflow=c2.has_outflow, build_solvers=False)
    assert o2.nsteps == 696                                      # BF_cav0.f00001 header: istep = 697
    assert abs(o2.bm1.sum() - 1.2) < 1e-12                       # [-0.5,0.5] x [0,1.2]
    lid = np.isclose(c2.y, 1.2) & (np.abs(c2.x) < 0.499)
    assert np.all(c2.ub[0][lid] == 1.0) and np.all(c2.mask[lid] == 0.0)


def test_cpu_port_matches_oracle(case6, oracle6, modes):
    """oracle/cpu_step.c (C + OpenMP, iterative solves: what bench.py times as cpu_baseline) against the numpy oracle
    (sparse direct solves) on the reference's eigenmode: two independent restatements of the same step."""
    from oracle.cpu_port import CpuPort
    o = oracle6
    cp = CpuPort(o, case6.meta["vert"], case6.meta["nvert"], tol_helm=1e-13, tol_pres=1e-13, tol_relative=0)
    u = modes["dRe_u"].astype(np.float64)
    q = (u[0], u[1], o.J12 @ modes["dRe_p"].astype(np.float64) @ o.J12.T)
    f = cp.matvec(q, nsteps=4)
    ref = o.matvec(q, nsteps=4)
    num = sum(np.sum(o.bm1 * (a - b) ** 2) for a, b in zip(f[:2], ref[:2]))
    den = sum(np.sum(o.bm1 * b ** 2) for b in ref[:2])
    assert np.sqrt(num / den) < 2e-9, np.sqrt(num / den)
    assert np.abs(f[2] - ref[2]).max() / np.abs(ref[2]).max() < 1e-5
    assert cp.stats["unconverged"] == 0 and cp.stats["steps"] == 4
    # thread count does not change the answer beyond reduction-order rounding
    cp.set_threads(1)
    f1 = cp.matvec(q, nsteps=2)
    cp.set_threads(cp.max_threads())
    cp.set_threads(4)
    f4 = cp.matvec(q, nsteps=2)
    assert max(np.abs(a - b).max() for a, b in zip(f1[:2], f4[:2])) < 1e-12


def test_cpu_port_projection_space_same_map_fewer_iterations(case6, oracle6, modes):
    """The pressure projection space of the C port (what bench.py's cpu_baseline runs since round 5, as the HIP path does):
    the same map as without it to the pressure tolerance, with fewer GMRES iterations per step; kept from map to map."""
    from oracle.cpu_port import CpuPort
    o = oracle6
    u = modes["dRe_u"].astype(np.float64)
    q = (u[0], u[1], o.J12 @ modes["dRe_p"].astype(np.float64) @ o.J12.T)
    kw = dict(tol_helm=1e-12, tol_pres=3e-2, tol_relative=1, min_pres=2)          # bench.py's pressure settings
    base = CpuPort(o, case6.meta["vert"], case6.meta["nvert"], **kw)
    f0 = base.matvec(q, nsteps=20)
    it0 = base.stats["pres_iters"]
    cp = CpuPort(o, case6.meta["vert"], case6.meta["nvert"], nproj=8, **kw)
    f1 = cp.matvec(q, nsteps=20)
    it1 = cp.stats["pres_iters"]
    num = sum(np.sum(o.bm1 * (a - b) ** 2) for a, b in zip(f1[:2], f0[:2]))
    den = sum(np.sum(o.bm1 * b ** 2) for b in f0[:2])
    print("relative difference of the maps %.2e, pressure iterations %d -> %d" % (np.sqrt(num / den), it0, it1))
    assert np.sqrt(num / den) < 1e-5, np.sqrt(num / den)
    assert cp.stats["unconverged"] == 0 and it1 < 0.8 * it0, (it0, it1)
    cp.matvec(q, nsteps=20)                           # the space of the last map is still there (another iteration history)
    assert cp.stats["pres_iters"] != it1 and cp.stats["pres_iters"] < 0.8 * it0
    cp.proj_reset()
    f3 = cp.matvec(q, nsteps=20)
    assert cp.stats["pres_iters"] == it1
    assert max(np.abs(a - b).max() for a, b in zip(f3[:2], f1[:2])) < 1e-12


def test_roofline_does_not_count_arrays_that_vanish():
    """nekstab_amd/roofline.py: metric terms, G factors and base-flow constants flagged in nsk_stats.zero_arrays are not inputs of
    the kernels (hexahedra only): the per-step bytes shrink by exactly those arrays."""
    from nekstab_amd import roofline
    geom = dict(nel=1000, lx1=8, ndim=3, nvert=1331, nproj=8, helm_iters=4.0, pres_iters=10.0, pres_jsum=45.0, coarse_bytes=1e6)
    full = roofline.per_step_bytes(**geom)
    za = sum(1 << b for b in (2, 5, 6, 7)) | (1 << 10) | (1 << 11) | sum(1 << (12 + b) for b in (2, 5, 8, 9, 10, 11))
    cut = roofline.per_step_bytes(zero_arrays=za, **geom)
    P, P2, Pd = 1000 * 512, 1000 * 216, 1000 * 1728
    assert full["K1 convect"] - cut["K1 convect"] == 8.0 * Pd * 6
    assert full["K3 helm iteration (x n_helm x d)"] - cut["K3 helm iteration (x n_helm x d)"] == 8.0 * 2 * P * 3 * 4.0
    assert full["K7 divgs (x n_pres)"] - cut["K7 divgs (x n_pres)"] == 8.0 * P2 * 4 * 10.0
    assert roofline.per_step_bytes(zero_arrays=za, **dict(geom, ndim=2, lx1=8)) == roofline.per_step_bytes(**dict(geom, ndim=2, lx1=8))
    rule, distinct = roofline.helm_launch_bytes(nel=1000, lx1=8, ndim=3, zero_arrays=za)
    rule0, distinct0 = roofline.helm_launch_bytes(nel=1000, lx1=8, ndim=3)
    assert rule0 - rule == 8.0 * 2 * P * 3 and distinct0 - distinct == 8.0 * 2 * P
