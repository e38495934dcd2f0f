"""Hexahedral (3-D) hot path through the C-ABI against the 3-D oracle (BASELINE configs 4 and 5 are 3-D):
element kernels, Helmholtz and pressure solves, direct / adjoint / nonlinear steps on small deformed boxes,
and the z-extruded cylinder against the 2-D GPU path."""
import os

import numpy as np
import pytest

from nekstab_amd import mesh3d

pytestmark = pytest.mark.gpu

TOL = dict(tol_helm=1e-12, tol_pres=1e-8, tol_relative=1, max_helm_iter=200, max_pres_iter=48)


def _ubf(x, y, z):
    return np.stack([1.0 - 0.3 * y * y + 0.1 * np.sin(x + z), 0.2 * np.cos(x) * y + 0.1 * z, 0.15 * np.sin(y + 0.5 * z) + 0.05 * x])


def _case(lx1, outflow):
    if outflow:
        # (lx1 = 10: 2 x 2 x 2 elements -- the oracle's sparse factorisations at that order cost 20 s on the 3 x 2 x 2 box)
        return mesh3d.box_case_3d(2 if lx1 >= 10 else 3, 2, 2, lx1, lengths=(2.0, 1.0, 0.8), outflow_xmax=True, re=40.0, endtime=0.05,
                                  ub_func=_ubf, warp=0.06)
    # closed box (pressure null space, `ortho`): undeformed, so that D^T 1 vanishes exactly on the free nodes and the
    # discrete E is exactly singular -- on curved elements it is only nearly so (Gauss quadrature of grad(phi_i)) and
    # a pinned direct solve and an iterative solve of the nearly singular system legitimately differ
    return mesh3d.box_case_3d(2, 2, 2, lx1, lengths=(1.0, 1.2, 0.9), re=30.0, endtime=0.05, ub_func=_ubf, warp=0.0,
                              stretch=lambda t: 0.5 * (1.0 - np.cos(np.pi * t)))


def _oracle(c, solvers=True):
    from oracle.linns3d import LinNS3D
    return LinNS3D(x=c.x, y=c.y, z=c.z, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re,
                   endtime=c.endtime, has_outflow=c.has_outflow, build_solvers=solvers)


def _hip(c, **kw):
    from nekstab_amd.capi import NekStabHip
    a = dict(TOL); a.update(kw)
    return NekStabHip(c, c.meta["vert"], c.meta["nvert"], **a)


@pytest.fixture(scope="module", params=[(6, True), (8, True), (6, False), (10, True)],
                ids=["lx6-outflow", "lx8-outflow", "lx6-closed", "lx10-outflow"])
def setup3(request):
    lx1, outflow = request.param
    c = _case(lx1, outflow)
    c.spng = 0.4 * np.clip(c.x - 1.4, 0.0, None) ** 2 if outflow else np.zeros_like(c.x)
    o = _oracle(c)
    h = _hip(c)
    yield c, o, h
    h.close()


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max()


def test_setup_matches(setup3):
    c, o, h = setup3
    assert h.nsteps == o.nsteps and abs(h.dt - o.dt) < 1e-15
    assert h.nvel == c.nel * c.lx1 ** 3 and h.npres == c.nel * (c.lx1 - 2) ** 3 and h.nstate == 3 * h.nvel + h.npres


def test_element_operators(setup3):
    c, o, h = setup3
    rng = np.random.default_rng(1)
    u = rng.standard_normal((3,) + c.x.shape)
    p = rng.standard_normal((c.nel,) + (c.lx1 - 2,) * 3)
    assert _rel(h.t_axhelm(u[0], 0.7, 1.3), o.axhelm(u[0], 0.7, 1.3)) < 1e-12
    assert _rel(h.t_dssum(u[1]), o.dssum(u[1])) < 1e-13
    assert _rel(h.t_op3(1, u), o.opdiv(u)) < 1e-12
    assert _rel(h.t_op3(2, p), np.stack(o.opgradt(p))) < 1e-12
    # E = D B^-1 D^T against the oracle's assembled sparse matrix
    assert _rel(h.t_eapply(p).ravel(), o._Emat @ p.ravel()) < 1e-11


def test_convection_terms(setup3):
    c, o, h = setup3
    rng = np.random.default_rng(2)
    u = rng.standard_normal((3,) + c.x.shape)
    U = o.ub
    sp = o.spng * o.bm1
    direct = np.stack([-(sp * u[k] + o.convect(u, U[k]) + o.convect(U, u[k])) for k in range(3)])
    assert _rel(h.t_op3(3, u, 0), direct) < 1e-12
    a = o.convect_adj(u, U)
    adj = np.stack([-(sp * u[k] + a[k] - o.convect(U, u[k])) for k in range(3)])
    assert _rel(h.t_op3(3, u, 1), adj) < 1e-12
    nl = np.stack([-o.convect(u, u[k]) for k in range(3)])
    assert _rel(h.t_op3(3, u, 2), nl) < 1e-12
    if c.lx1 in (8, 10):
        # the same terms with the tensor contractions on v_mfma_f64_16x16x4_f64 (nsk3_mfma.hpp: k_convect_mfma8, k_convect_mfma<10>):
        # against the oracle and against the thread-per-node kernel
        assert _rel(h.t_op3(8, u, 0), direct) < 1e-12
        assert _rel(h.t_op3(8, u, 1), adj) < 1e-12
        assert _rel(h.t_op3(8, u, 0), h.t_op3(3, u, 0)) < 1e-13
    if c.lx1 == 10:
        # the full equations' term on the matrix cores (k_convect_mfma_nl<10>: Newton-Krylov at config 5's order)
        assert _rel(h.t_op3(8, u, 2), nl) < 1e-12
        assert _rel(h.t_op3(8, u, 2), h.t_op3(3, u, 2)) < 1e-13


def test_helmholtz_solve(setup3):
    c, o, h = setup3
    rng = np.random.default_rng(3)
    r = rng.standard_normal((3,) + c.x.shape)
    out, iters = h.t_op3(5, r, 3)
    h2 = (11.0 / 6.0) / o.dt
    ref = np.stack([o.helm_solve(r[k], o.nu, h2) for k in range(3)])
    assert _rel(out, ref) < 1e-9
    assert 0 < iters < 200


def test_pressure_solve(setup3):
    c, o, h = setup3
    rng = np.random.default_rng(4)
    g = rng.standard_normal((c.nel,) + (c.lx1 - 2,) * 3)
    if not c.has_outflow:
        g -= g.mean()
    x, iters = h.t_pres_solve(g)
    ref = o.E_solve(g)
    if not c.has_outflow:
        x = x - x.mean()
    assert _rel(x, ref) < 1e-6, iters
    assert 0 < iters <= 48


@pytest.mark.parametrize("mode", [0, 1])
def test_steps_match_oracle(setup3, mode):
    c, o, h = setup3
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask]
    m = c.lx1 - 2
    p = np.zeros((c.nel, m, m, m))
    nst = 5
    ref = o.matvec((q[0], q[1], q[2], p), adjoint=bool(mode), nsteps=nst)
    v0, v1 = h.alloc(2)
    h.upload3(v0, q[0], q[1], q[2], p)
    keep = h.nsteps
    h.set_nsteps(nst)
    h.matvec(v1, v0, mode)
    h.set_nsteps(keep)
    out = h.download3(v1)
    sc = max(np.abs(ref[k]).max() for k in range(3))
    for k in range(3):
        assert np.abs(out[k] - ref[k]).max() < 1e-7 * sc
    pr, po = ref[3], out[3]
    if not c.has_outflow:
        pr, po = pr - pr.mean(), po - po.mean()
    assert np.abs(po - pr).max() < 1e-4 * max(np.abs(pr).max(), 1e-30)
    st = h.stats()
    assert st["unconverged"] == 0
    h.free([v0, v1])


def test_krylov_algebra_3d(setup3):
    c, o, h = setup3
    rng = np.random.default_rng(5)
    a = [rng.standard_normal(c.x.shape) for _ in range(3)]
    b = [rng.standard_normal(c.x.shape) for _ in range(3)]
    m = c.lx1 - 2
    pa, pb = rng.standard_normal((c.nel, m, m, m)), rng.standard_normal((c.nel, m, m, m))
    va, vb = h.alloc(2)
    h.upload3(va, *a, pa); h.upload3(vb, *b, pb)
    assert abs(h.dot(va, vb) - o.inner(a, b)) < 1e-12 * abs(o.inner(a, a))
    h.axpy(va, -0.5, vb)
    out = h.download3(va)
    assert np.abs(out[2] - (a[2] - 0.5 * b[2])).max() < 1e-14 and np.abs(out[3] - (pa - 0.5 * pb)).max() < 1e-14
    h.free([va, vb])


def test_extruded_cylinder_matches_2d():
    """z-invariant perturbation on the z-extruded cylinder mesh (the reference's own geometry): the hexahedral
    path must reproduce the quadrilateral path plane by plane."""
    import os
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    here = os.path.dirname(os.path.abspath(__file__))
    c2 = mesh.load_case_npz(os.path.join(here, "golden", "cylinder_case.npz"), 6)
    modes = np.load(os.path.join(here, "golden", "cylinder_modes.npz"))
    c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
    prod = dict(tol_helm=1e-11, tol_pres=1e-4, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=8)
    h2 = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], **prod)
    h3 = _hip(c3, **prod)
    try:
        assert h3.nsteps == h2.nsteps
        u, v = modes["dRe_u"][0].astype(np.float64) * c2.mask, modes["dRe_u"][1].astype(np.float64) * c2.mask
        nst = 6
        a0, a1 = h2.alloc(2)
        h2.upload(a0, u, v, np.zeros(h2.npres)); h2.set_nsteps(nst); h2.matvec(a1, a0, 0)
        r2 = h2.download(a1)
        b0, b1 = h3.alloc(2)
        h3.upload3(b0, mesh3d.extrude_field(u, 2), mesh3d.extrude_field(v, 2), np.zeros(h3.nvel), np.zeros(h3.npres))
        h3.set_nsteps(nst); h3.matvec(b1, b0, 0)
        r3 = h3.download3(b1)
        sc = np.abs(r2[0]).max()
        for k in range(6):
            for layer in range(2):
                e = slice(layer * c2.nel, (layer + 1) * c2.nel)
                assert np.abs(r3[0][e, k] - r2[0]).max() < 1e-7 * sc
                assert np.abs(r3[1][e, k] - r2[1]).max() < 1e-7 * sc
        assert np.abs(r3[2]).max() < 1e-7 * sc
        assert h3.stats()["unconverged"] == 0
    finally:
        h2.close(); h3.close()


def test_ifbf2d_forces_the_third_base_flow_component_to_zero():
    """core/matvec.f:110-112: a hexahedral run about a two-dimensional base flow zeroes the base flow's vz before the
    linearised solver is prepared.  A context given a polluted third component with ifbf2d equals the context of the clean
    extrusion: same dt / nsteps, same map."""
    import dataclasses
    import os
    from nekstab_amd import mesh
    here = os.path.dirname(os.path.abspath(__file__))
    c2 = mesh.load_case_npz(os.path.join(here, "golden", "cylinder_case.npz"), 6)
    c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
    ub = c3.ub.copy()
    ub[2] = 0.3 * np.sin(c3.x) * c3.mask                       # what a restart file of a 3-D DNS may carry in vz
    dirty = dataclasses.replace(c3, ub=ub)
    kw = dict(tol_helm=1e-11, tol_pres=1e-5, tol_relative=1, max_helm_iter=150, max_pres_iter=144)
    ha, hb, hc = _hip(c3, **kw), _hip(dirty, ifbf2d=True, **kw), _hip(dirty, **kw)
    try:
        assert hb.nsteps == ha.nsteps and hb.dt == ha.dt
        x, y, z = c3.x, c3.y, c3.z
        q = [np.sin(x + np.pi * z) * c3.mask, np.cos(y) * np.cos(np.pi * z) * c3.mask, np.sin(x - y) * c3.mask, np.zeros(ha.npres)]
        out = []
        for h in (ha, hb, hc):
            v0, v1 = h.alloc(2)
            h.upload3(v0, *q); h.set_nsteps(3); h.matvec(v1, v0, 0)
            out.append(h.download3(v1))
        for k in range(3):
            assert np.array_equal(out[0][k], out[1][k])
        assert np.abs(out[2][0] - out[0][0]).max() > 1e-6 * np.abs(out[0][0]).max()     # (without the switch vz does act)
    finally:
        ha.close(); hb.close(); hc.close()


def test_arrays_that_are_zero_on_every_node_are_not_loaded_same_bits(monkeypatch):
    """Spanwise-extruded mesh, two-dimensional base flow (BASELINE config 4's shape): four of the nine metric terms, g5 and g6 and
    six of the twelve base-flow constants of the convection kernel vanish on every node.  The set-up finds them (stats:
    zero_arrays), the kernels skip their loads; with option zero_metrics = 0 every array is loaded again: bit-identical maps,
    direct and adjoint.  A context set up with NSK_ZERO_METRICS=0 (the rounding noise of the derivatives kept, as rounds 1-4)
    agrees to rounding."""
    import os
    from nekstab_amd import mesh
    here = os.path.dirname(os.path.abspath(__file__))
    c2 = mesh.load_case_npz(os.path.join(here, "golden", "cylinder_case.npz"), 8)
    c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
    kw = dict(tol_helm=1e-11, tol_pres=1e-8, tol_relative=1, max_helm_iter=150, max_pres_iter=144)
    x, y, z = c3.x, c3.y, c3.z
    q = [np.sin(x + np.pi * z) * c3.mask, np.cos(y) * np.cos(np.pi * z) * c3.mask, np.sin(x - y) * c3.mask, np.zeros(1)]
    ha = _hip(c3, **kw)
    monkeypatch.setenv("NSK_ZERO_METRICS", "0")
    hb = _hip(c3, **kw)
    monkeypatch.delenv("NSK_ZERO_METRICS")
    try:
        za, zb = ha.stats()["zero_arrays"], hb.stats()["zero_arrays"]
        metrics = {2, 5, 6, 7}; gfac = {10, 11}; bfc = {12 + 2, 12 + 5, 12 + 8, 12 + 9, 12 + 10, 12 + 11}
        assert {b for b in range(24) if za >> b & 1} == metrics | gfac | bfc, sorted(b for b in range(24) if za >> b & 1)
        assert zb == 0
        q[3] = np.zeros(ha.npres)
        out = {}
        for name, h, opt in (("masked", ha, 1), ("loaded", ha, 0), ("noise", hb, None)):
            if opt is not None:
                h.set_option("zero_metrics", opt)
                assert (h.stats()["zero_arrays"] != 0) == bool(opt)
            v0, v1 = h.alloc(2)
            h.upload3(v0, *q); h.set_nsteps(3)
            for adj in (0, 1):
                h.matvec(v1, v0, adj)
                out[name, adj] = h.download3(v1)
        for adj in (0, 1):
            for k in range(4):
                assert np.array_equal(out["masked", adj][k], out["loaded", adj][k])
                sc = max(np.abs(out["noise", adj][k]).max(), 1e-30)
                assert np.abs(out["masked", adj][k] - out["noise", adj][k]).max() < 1e-6 * sc
        assert ha.stats()["unconverged"] == 0
    finally:
        ha.close(); hb.close()


def test_a_three_dimensional_base_flow_on_an_extruded_mesh_takes_the_base_flow_bits_back():
    """nsk_set_baseflow re-scans the twelve base-flow constants: with a spanwise-varying base flow on the extruded mesh only the
    mapping's zeros stay in `zero_arrays`, and the map equals the one with every array loaded, bit for bit."""
    import os
    from nekstab_amd import mesh
    here = os.path.dirname(os.path.abspath(__file__))
    c2 = mesh.load_case_npz(os.path.join(here, "golden", "cylinder_case.npz"), 6)
    c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
    kw = dict(tol_helm=1e-11, tol_pres=1e-8, tol_relative=1, max_helm_iter=150, max_pres_iter=144)
    h = _hip(c3, **kw)
    try:
        assert (h.stats()["zero_arrays"] >> 12) != 0                       # two-dimensional base flow: six constants vanish
        x, y, z = c3.x, c3.y, c3.z
        b0, v0, v1 = h.alloc(3)
        h.upload3(b0, c3.ub[0] * (1.0 + 0.05 * np.cos(2 * np.pi * z)), c3.ub[1] + 0.05 * np.sin(2 * np.pi * z) * c3.mask, 0.1 * np.cos(2 * np.pi * z) * np.sin(x) * c3.mask, np.zeros(h.npres))
        h.set_baseflow(b0)
        za = h.stats()["zero_arrays"]
        assert (za >> 12) == 0 and {b for b in range(12) if za >> b & 1} == {2, 5, 6, 7, 10, 11}
        q = [np.sin(x + np.pi * z) * c3.mask, np.cos(y) * np.cos(np.pi * z) * c3.mask, np.sin(x - y) * c3.mask, np.zeros(h.npres)]
        h.upload3(v0, *q); h.set_nsteps(3)
        h.matvec(v1, v0, 0); a = h.download3(v1)
        h.set_option("zero_metrics", 0)
        assert h.stats()["zero_arrays"] == 0
        h.matvec(v1, v0, 0); b = h.download3(v1)
        for k in range(4):
            assert np.array_equal(a[k], b[k])
        assert h.stats()["unconverged"] == 0
    finally:
        h.close()


def test_host_checked_convergence_full_mesh_context():
    """Option hostcheck (default on hexahedral meshes of >= 8192 elements, where a launch that only finds its solve converged
    costs 25-140 us and a map redone with larger launch budgets tens of seconds): eager steps, the host reads the device's
    convergence flags and stops issuing iterations.  Bit-identical to the budgeted launches (captured or eager), same counts,
    and no launch budget to exceed."""
    c = _case(8, True)
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel, 6, 6, 6))]
    out, its = {}, {}
    for name, opts in (("graph", {}), ("eager", {"use_graph": 0}), ("hostcheck", {"hostcheck": 1})):
        h = _hip(c, nproj=4)
        try:
            for k, v in opts.items():
                h.set_option(k, v)
            a, b = h.alloc(2)
            h.upload3(a, *q)
            h.set_nsteps(5)
            res = []
            for rep in range(2):
                h.matvec(b, a, 0)
                res.append(h.download3(b))
                h.copy(a, b)
            out[name] = res
            st = h.stats()
            its[name] = (st["helm_iters"], st["pres_iters"], st["retries"])
        finally:
            h.close()
    print("iterations of the last map (velocity, pressure, redone maps):", its)
    for name in ("eager", "hostcheck"):
        for rep in range(2):
            for x0, x1 in zip(out["graph"][rep], out[name][rep]):
                assert np.array_equal(x0, x1), (name, rep)
        assert its[name][:2] == its["graph"][:2]
    assert its["hostcheck"][2] == 0


def test_arnoldi_3d_matches_oracle():
    """Short Arnoldi factorisation on the hexahedral path (full-length maps, host loop of krylov.py) against
    the oracle's Arnoldi (reference algorithm: core/krylov_decomposition.f:7-202)."""
    from nekstab_amd import krylov
    from oracle.linns import arnoldi
    c = _case(6, True)
    c.spng = 0.4 * np.clip(c.x - 1.4, 0.0, None) ** 2
    o = _oracle(c)
    h = _hip(c)
    try:
        x, y, z = c.x, c.y, c.z
        q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
             np.sin(x + y) * np.cos(2.0 * z) * c.mask]
        p = np.zeros((c.nel, 4, 4, 4))
        kdim = 5
        _, Href = arnoldi(o, (q[0], q[1], q[2], p), kdim)
        v0 = h.alloc(1)[0]
        h.upload3(v0, q[0], q[1], q[2], p)
        res = krylov.krylov_schur(h, v0, kdim, mode=0, schur_tgt=0)
        assert np.abs(res.H - Href).max() < 2e-7 * np.abs(Href).max()
        vr = np.sort_complex(np.linalg.eigvals(Href[:kdim, :kdim]))
        vg = np.sort_complex(res.vals)
        assert np.abs(vr - vg).max() < 1e-6
    finally:
        h.close()


@pytest.mark.parametrize("lx1", [6, 10])
def test_nonlinear_map_and_relinearisation_3d(lx1):
    """Full-equation steps (newton_krylov's nonlinear map) and a new linearisation point on hexahedra (lx1 = 10: the
    convection kernel with its accumulators in LDS, both of its branches)."""
    if lx1 == 10:          # (two sparse-LU oracles at this order: a 2 x 2 x 1 box keeps them at seconds instead of 2 x 20 s)
        c = mesh3d.box_case_3d(2, 2, 1, lx1, lengths=(1.4, 1.0, 0.5), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=_ubf, warp=0.06)
        c.spng = 0.4 * np.clip(c.x - 0.9, 0.0, None) ** 2
    else:
        c = _case(lx1, True)
    o = _oracle(c)
    h = _hip(c)
    try:
        m = c.lx1 - 2
        q = [c.ub[0].copy(), c.ub[1].copy(), c.ub[2].copy(), np.zeros((c.nel, m, m, m))]
        nst = 4
        ref = o.nonlinear_map(q, nsteps=nst)
        v0, v1, v2 = h.alloc(3)
        h.upload3(v0, *q)
        h.set_nsteps(nst)
        h.nonlinear_map(v1, v0)
        out = h.download3(v1)
        sc = max(np.abs(ref[k]).max() for k in range(3))
        for k in range(3):
            assert np.abs(out[k] - ref[k]).max() < 1e-7 * sc
        # linearise about the advanced state: dt / nsteps follow the CFL rule of the new base flow
        from oracle.linns3d import LinNS3D
        o2 = LinNS3D(x=c.x, y=c.y, z=c.z, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=np.stack(ref[:3]), spng=c.spng,
                     re=c.re, endtime=c.endtime, has_outflow=c.has_outflow)
        h.set_baseflow(v1)
        assert h.nsteps == o2.nsteps and abs(h.dt - o2.dt) < 1e-14
        x, y, z = c.x, c.y, c.z
        pert = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
                np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel, m, m, m))]
        r2 = o2.matvec(pert, nsteps=3)
        h.upload3(v0, *pert)
        h.set_nsteps(3)
        h.matvec(v2, v0, 0)
        g2 = h.download3(v2)
        sc = max(np.abs(r2[k]).max() for k in range(3))
        for k in range(3):
            assert np.abs(g2[k] - r2[k]).max() < 1e-7 * sc
    finally:
        h.close()


def _cavity_2d(re):
    import os
    from nekstab_amd import mesh, nekio
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "cavity_case.npz"))
    bcs = [(int(a), int(b), np.zeros(5), str(cd)) for (a, b), cd in zip(z["bc_ef"], z["bc_code"])]
    m = nekio.Re2Mesh(2, z["xc"].shape[0], z["xc"], z["yc"], None, [], bcs)
    c2 = mesh.build_case_2d(m, z["vlex"].astype(np.int64), z["bf_u"].astype(np.float64), 6, re=re, endtime=1.0, spng_str=0.0)
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    return c2, J @ z["bf_p"].astype(np.float64) @ J.T


def test_lid_driven_cavity_baseflow_is_a_fixed_point_2d_and_3d():
    """Reference data pin for the closed-cavity (pressure null space) nonlinear path: the committed base flow of
    examples/lid_driven (BF_cav0.f00001: 100 elements, lx1=6, y rescaled to [0,1.2], header istep = 697) is a fixed
    point of Phi_T at Re = 3600 (viscosity = -3600, cav.par:31) and at no other Reynolds number; its z-extrusion is
    the same fixed point of the hexahedral map."""
    from nekstab_amd.capi import NekStabHip
    kw = dict(tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=0, max_helm_iter=150, max_pres_iter=48)
    res = {}
    for re in (3600.0, 4000.0):
        c2, p2 = _cavity_2d(re)
        h = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], **kw)
        assert h.nsteps == 696                      # file header: istep = nsteps + 1 = 697  (third dt-rule pin)
        q, f = h.alloc(2)
        h.upload(q, c2.ub[0], c2.ub[1], p2)
        h.nonlinear_map(f, q, subtract_q=True)
        res[re] = h.norm(f) ** 2
        h.close()
    assert res[3600.0] < 2e-9 and res[4000.0] > 1e-6, res
    c2, p2 = _cavity_2d(3600.0)
    c3 = mesh3d.extrude_case(c2, 2, 0.4, periodic=True)
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], **kw)
    try:
        q, f = h.alloc(2)
        h.upload3(q, mesh3d.extrude_field(c2.ub[0], 2), mesh3d.extrude_field(c2.ub[1], 2), np.zeros(c3.x.shape),
                  mesh3d.extrude_pressure(p2, 2))
        h.nonlinear_map(f, q, subtract_q=True)
        r3 = h.norm(f) ** 2 / 0.4                   # the 3-D norm integrates over the span
        assert abs(r3 - res[3600.0]) < 1e-3 * res[3600.0], (r3, res)
        assert h.stats()["unconverged"] == 0
    finally:
        h.close()


def test_newton_krylov_3d_box():
    """Newton-Krylov (core/newton_krylov.f:5-296 restated in newton.py) on a hexahedral context: lid-driven box,
    spanwise periodic, Re = 50; the converged state must be a fixed point of the 3-D oracle's map as well."""
    from nekstab_amd import newton
    lid = lambda x, y, z: np.stack([np.where(np.isclose(y, 1.0), (1.0 - (2 * x - 1.0) ** 2) ** 2 * (1.0 + 0.3 * np.sin(2 * np.pi * z / 0.6)), 0.0),
                                    0.0 * x, 0.0 * x])
    c = mesh3d.box_case_3d(3, 3, 2, 6, lengths=(1.0, 1.0, 0.6), periodic=(False, False, True), re=50.0, endtime=0.5, ub_func=lid)
    h = _hip(c, tol_pres=1e-7)
    try:
        m = c.lx1 - 2
        q = h.alloc(1)[0]
        h.upload3(q, c.ub[0], c.ub[1], c.ub[2], np.zeros((c.nel, m, m, m)))
        its, hist = newton.newton_krylov(h, q, k_dim=30, tol=1e-12, maxiter_newton=8)
        assert hist[-1] < 1e-12 and its <= 8, hist
        out = h.download3(q)
        assert np.abs(out[2]).max() > 1e-4          # genuinely three-dimensional
        o = _oracle(c)
        o.ub = np.stack(out[:3]); o.dt, o.nsteps = o.timestep_rule()
        assert o.nsteps == h.nsteps
        ref = o.nonlinear_map(out)
        sc = max(np.abs(out[k]).max() for k in range(3))
        for k in range(3):
            assert np.abs(ref[k] - out[k]).max() < 5e-6 * sc
    finally:
        h.close()


def test_field_files_and_checkpoint_3d(tmp_path):
    """Krylov vectors / eigenmodes of a hexahedral run in the reference's field-file format (3-D #std files, pressure
    on mesh 1) and back (core/eigensolvers.f:802-905, core/IO.f:15-60)."""
    from nekstab_amd import checkpoint, nekio
    c = _case(6, True)
    h = _hip(c)
    try:
        rng = np.random.default_rng(7)
        m = c.lx1 - 2
        a = [rng.standard_normal(c.x.shape) * c.mask for _ in range(3)]
        J12 = checkpoint._maps(c)[1]
        p1 = np.sin(c.x + 2 * c.y - c.z)                           # a mesh-1 pressure that mesh 2 represents
        p2 = np.einsum("ci,bj,ak,ekji->ecba", J12, J12, J12, p1, optimize=True)
        v, w = h.alloc(2)
        h.upload3(v, *a, p2)
        path = str(tmp_path / checkpoint.kry_name("box", 3))
        checkpoint.write_krylov_vector(h, c, v, path, time=2.0)
        f = nekio.read_fld(path)
        assert (f.nx, f.ny, f.nz, f.nel) == (6, 6, 6, c.nel) and f.u.shape == (3, c.nel, 6, 6, 6) and f.istep == h.nsteps + 1
        assert np.abs(f.x[2] - c.z).max() == 0.0
        checkpoint.read_krylov_vector(h, c, w, path)
        out = h.download3(w)
        for k in range(3):
            assert np.abs(out[k] - a[k]).max() == 0.0
        assert np.abs(out[3] - p2).max() < 1e-12 * np.abs(p2).max()   # mesh 2 -> 1 -> 2 is the identity on P_{lx2-1}
    finally:
        h.close()


def test_arnoldi_on_extruded_cylinder_finds_the_reference_eigenvalue(spectre):
    """End to end on hexahedra: k_dim = 90 Arnoldi on the cylinder mesh extruded in z (3992 elements, lx1 = 6) from a
    fully three-dimensional noise seed.  The spanwise-periodic spectrum contains the 2-D one and at Re = 50 its leading
    pair is the 2-D Hopf pair, i.e. row 1 of the reference's Spectre_Hd.dat (0.7387113 + 0.6972442i)."""
    import os
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.capi import NekStabHip
    here = os.path.dirname(os.path.abspath(__file__))
    c2 = mesh.load_case_npz(os.path.join(here, "golden", "cylinder_case.npz"), 6)
    c3 = mesh3d.extrude_case(c2, 2, 2.0, periodic=True)
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8,
                   max_helm_iter=150, max_pres_iter=48)
    try:
        qx, qy = seed.add_noise(c2)
        z = c3.z
        ex = lambda f: mesh3d.extrude_field(f, 2)
        v0 = h.alloc(1)[0]
        h.upload3(v0, ex(qx) * (1 + 0.3 * np.sin(np.pi * z)) * c3.mask, ex(qy) * (1 + 0.3 * np.cos(np.pi * z)) * c3.mask,
                  0.3 * ex(qx) * np.sin(np.pi * z) * c3.mask, np.zeros(h.npres))
        res = krylov.krylov_schur(h, v0, 90, mode=0, schur_tgt=0)
        ref = spectre["Hd"][0]
        mu = res.vals[0] if res.vals[0].imag > 0 else res.vals[1]
        assert res.residual[0] < 2e-5
        assert abs(mu - complex(ref[0], abs(ref[1]))) < 5e-6, mu
    finally:
        h.close()


def test_chebyshev_coarse_solve_matches_dense_inverse(monkeypatch):
    """Large coarse spaces (configs 4 and 5: 5e4-1e5 vertices) use a sparse A_c with a fixed-degree Chebyshev-Jacobi
    polynomial instead of the dense inverse; forced here on a small mesh it must give the same pressure solution and
    (nearly) the same GMRES iteration count."""
    c = _case(6, True)
    o = _oracle(c)
    rng = np.random.default_rng(4)
    g = rng.standard_normal((c.nel, 4, 4, 4))
    ref = o.E_solve(g)
    h = _hip(c)
    x0, it0 = h.t_pres_solve(g)
    h.close()
    monkeypatch.setenv("NSK_COARSE_ITER", "1")
    h = _hip(c)
    try:
        x1, it1 = h.t_pres_solve(g)
        assert _rel(x1, ref) < 1e-6 and _rel(x0, ref) < 1e-6
        assert 0 < it1 <= it0 + 3, (it0, it1)
    finally:
        h.close()


def test_block_circulant_coarse_solve_equals_dense_inverse(monkeypatch):
    """Meshes made of uniform periodic layers (BASELINE config 4: 30 spanwise layers) have a block-circulant vertex operator:
    the coarse solve is exact through a real Fourier transform in z and one stored inverse per wavenumber (nsk3_setup.inc),
    three launches instead of the Chebyshev polynomial's dozens.  Forced on a small extrusion (closed cavity: the singular
    wavenumber-0 block included; and the cylinder with its outflow) it must reproduce the dense inverse's pressure solution
    AND its GMRES iteration count (both coarse solves are exact)."""
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from tests.conftest import GOLDEN
    c2a, _ = _cavity_2d(3600.0)
    c2b = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    for c2, nz, lz in ((c2a, 5, 1.0), (c2b, 4, 2.0)):
        c3 = mesh3d.extrude_case(c2, nz, lz, periodic=True)
        rng = np.random.default_rng(4)
        g = rng.standard_normal((c3.nel, 4, 4, 4))
        out = {}
        for name, env in (("dense", {"NSK_COARSE_ITER": "0", "NSK_COARSE_CIRC": "0"}), ("circulant", {"NSK_COARSE_ITER": "0", "NSK_COARSE_CIRC": "1"}),
                          ("chebyshev", {"NSK_COARSE_ITER": "1", "NSK_COARSE_CIRC": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=1e-8, tol_relative=1, max_helm_iter=200, max_pres_iter=48)
            try:
                out[name] = h.t_pres_solve(g)
            finally:
                h.close()
        xd, itd = out["dense"]; xc, itc = out["circulant"]; xp, itp = out["chebyshev"]
        print("extrusion of %d elements over %d layers: GMRES iterations dense %d, block-circulant %d, Chebyshev polynomial %d" % (c2.nel, nz, itd, itc, itp))
        assert abs(itc - itd) <= 1, (itd, itc)
        sc = np.abs(xd - xd.mean()).max()
        assert np.abs((xc - xc.mean()) - (xd - xd.mean())).max() < 1e-5 * sc


@pytest.mark.parametrize("lx1,outflow", [(6, True), (8, True), (8, False), (10, True)])
def test_lagged_gram_schmidt_equals_classic_two_pass(lx1, outflow):
    """Option gs_lag (default on single-rank hexahedral contexts): the second Gram-Schmidt correction of a GMRES basis vector
    is applied by the NEXT column's streaming pass instead of a pass of its own (two basis reads per column instead of four).
    Same Krylov method: a pressure solve to 1e-8 gives the classic sequence's solution and iteration count (+-1), short
    restart cycles included; gs_lag = 2 (first-pass dots in their own streaming kernel) likewise; a 5-step map agrees to
    solver tolerance."""
    c = _case(lx1, outflow)
    c.spng = np.zeros_like(c.x)
    rng = np.random.default_rng(11)
    g = rng.standard_normal((c.nel,) + (c.lx1 - 2,) * 3)
    if not outflow:
        g -= g.mean()
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel,) + (c.lx1 - 2,) * 3)]
    sol, its, maps = {}, {}, {}
    for lag in (0, 1, 2):
        for cyc in (48, 6):
            h = _hip(c, max_pres_iter=96)
            try:
                h.set_option("gs_lag", lag)
                h.set_option("gmres_cycle", cyc)
                xs, it = h.t_pres_solve(g)
                sol[lag, cyc], its[lag, cyc] = (xs if outflow else xs - xs.mean()), it
                if cyc == 48:
                    a, b = h.alloc(2)
                    h.upload3(a, *q)
                    h.set_nsteps(5)
                    h.matvec(b, a, 1)
                    maps[lag] = h.download3(b)
                    assert h.stats()["unconverged"] == 0
            finally:
                h.close()
    print("GMRES iterations (gs_lag, cycle):", its)
    for lag in (1, 2):
        for cyc in (48, 6):
            assert _rel(sol[lag, cyc], sol[0, cyc]) < 2e-6, (lag, cyc)
            assert abs(its[lag, cyc] - its[0, cyc]) <= (1 if cyc == 48 else 3), (lag, cyc, its)
        sc = max(np.abs(maps[0][k]).max() for k in range(3))
        for k in range(3):
            assert np.abs(maps[lag][k] - maps[0][k]).max() < 1e-7 * sc


@pytest.mark.parametrize("lx1,outflow", [(6, True), (8, True), (8, False), (10, True)])
def test_schwarz_kernel_forms_agree(lx1, outflow):
    """Option eapply_pipe: the Schwarz + D^T kernel of the pressure iteration as one workgroup per element (0: k_schwarz),
    as resident workgroups with the next element's inputs in flight (1: k_schwarz_p) and as one wavefront per element with
    no workgroup barrier (2: k_schwarz_w; 4: k_schwarz_w16, the default at lx1 <= 8 -- the solve in place in one tile, sixteen
    elements per CU in flight; 3: the divergence kernel in that form too, k_divgs_w: built, measured slower, not the default;
    5: four wavefronts per element, k_schwarz_q -- lx1 = 10 only, the default form elsewhere).
    Same arithmetic except for the order of the sums inside the
    matrix-core passes (forms 1 and 2 run the six fast-diagonalisation passes there too): a pressure solve to 1e-8 and a
    five-step adjoint map agree to rounding amplified by the solves, with the same iteration counts."""
    c = _case(lx1, outflow)
    c.spng = np.zeros_like(c.x)
    rng = np.random.default_rng(5)
    g = rng.standard_normal((c.nel,) + (c.lx1 - 2,) * 3)
    if not outflow:
        g -= g.mean()
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel,) + (c.lx1 - 2,) * 3)]
    sol, its, maps = {}, {}, {}
    for form in (0, 1, 2, 3, 4, 5):
        h = _hip(c, max_pres_iter=96)
        try:
            h.set_option("eapply_pipe", form)
            h.set_option("divgs_c3", 1 if form == 1 else 0)          # (the divergence kernel with its components side by side rides along with form 1)
            xs, it = h.t_pres_solve(g)
            sol[form], its[form] = (xs if outflow else xs - xs.mean()), it
            a, b = h.alloc(2)
            h.upload3(a, *q)
            h.set_nsteps(5)
            h.matvec(b, a, 1)
            maps[form] = h.download3(b)
            assert h.stats()["unconverged"] == 0
        finally:
            h.close()
    print("GMRES iterations by kernel form:", its)
    sc = max(np.abs(maps[0][k]).max() for k in range(3))
    for form in (1, 2, 3, 4, 5):
        assert _rel(sol[form], sol[0]) < 1e-7, form
        assert abs(its[form] - its[0]) <= 1, its
        for k in range(3):
            assert np.abs(maps[form][k] - maps[0][k]).max() < 1e-9 * sc, (form, k)


def test_resident_helmholtz_iteration_is_bit_identical_lx1_10():
    """Option helm_pf (default at lx1 = 10): the CG iteration as resident workgroups that walk their elements with the next
    element's r, p, s, x arriving in LDS by LDS-DMA (k_helm_p) against one workgroup per element (k_helm<10>).  Same arithmetic
    in the same order: a five-step direct map and the iteration counts are equal BIT FOR BIT, with eight resident workgroups
    (every workgroup walks several elements: prologue, steady state and last element of the loop) and with the default grid."""
    c = mesh3d.box_case_3d(4, 3, 3, 10, lengths=(2.0, 1.0, 0.8), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=_ubf, warp=0.06)      # 36 elements
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel,) + (c.lx1 - 2,) * 3)]
    out, its = {}, {}
    for form in ("wg", "resident8", "resident"):
        h = _hip(c)
        try:
            h.set_option("helm_pf", 0 if form == "wg" else 1)
            if form == "resident8":
                assert c.nel > 16
                h.set_option("helm_pf_grid", 8)
            a, b = h.alloc(2)
            h.upload3(a, *q)
            h.set_nsteps(5)
            h.matvec(b, a, 0)
            out[form] = h.download3(b)
            st = h.stats()
            its[form] = (st["helm_iters"], st["pres_iters"])
            assert st["unconverged"] == 0
        finally:
            h.close()
    for form in ("resident8", "resident"):
        assert its[form] == its["wg"], its
        for k in range(4):
            assert np.array_equal(out[form][k], out["wg"][k]), (form, k, np.abs(out[form][k] - out["wg"][k]).max())


@pytest.mark.parametrize("lx1,outflow", [(8, True), (6, False), (10, True)])
def test_streaming_projection_kernels_equal_element_kernel_forms(lx1, outflow):
    """Option flat_proj (default on single-rank hexahedral contexts): the once-per-step sums over the GMRES basis and the
    pressure projection space run as streaming kernels (k_pres_comb + k_gradt, k_proj_dots) instead of inside the element
    kernels k_pres_update / k_vel_update_proj.  Same arithmetic in another summation order: three consecutive maps with the
    projection space filling up agree to rounding-level differences amplified by the solves, with the same iteration counts
    (+-2), short GMRES restart cycles included."""
    c = _case(lx1, outflow)
    c.spng = np.zeros_like(c.x)
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel,) + (c.lx1 - 2,) * 3)]
    out, its = {}, {}
    for flat in (0, 1):
        for cyc in (48, 5):
            h = _hip(c, nproj=6, max_pres_iter=96)
            try:
                h.set_option("flat_proj", flat)
                h.set_option("gmres_cycle", cyc)
                a, b = h.alloc(2)
                h.upload3(a, *q)
                h.set_nsteps(4)
                res, cnt = [], []
                for rep in range(3):
                    h.matvec(b, a, 0)
                    res.append(h.download3(b))
                    st = h.stats()
                    assert st["unconverged"] == 0
                    cnt.append(st["pres_iters"])
                    h.copy(a, b)
                out[flat, cyc], its[flat, cyc] = res, cnt
            finally:
                h.close()
    print("pressure iterations per map (flat_proj, cycle):", its)
    for cyc in (48, 5):
        for rep in range(3):
            sc = max(np.abs(out[0, cyc][rep][k]).max() for k in range(3))
            for k in range(3):
                assert np.abs(out[1, cyc][rep][k] - out[0, cyc][rep][k]).max() < 2e-7 * sc, (cyc, rep, k)
            assert abs(its[1, cyc][rep] - its[0, cyc][rep]) <= 2, its


@pytest.mark.parametrize("lx1", [6, 8])
def test_block_fdm_velocity_preconditioner_on_a_stretched_box(lx1, monkeypatch):
    """Option helm_fdm (built, measured, NOT a gain -- DESIGN.md section 7): velocity solves preconditioned by
    element-block fast diagonalisation (k_helm_fa / k_helm_fb: two launches per CG iteration) instead of the Jacobi diagonal.
    What is tested is that it is a correct preconditioned CG: on a closed box with Chebyshev-like clustering the solve reproduces
    the oracle's direct solve and a 4-step map equals the Jacobi path's at solver tolerance; the iteration counts are printed.
    The library never chooses it by itself."""
    stretch = lambda t: 0.5 * (1.0 - np.cos(np.pi * t))
    # (lx1 = 8 on a smaller box, and the oracle without its pressure factorisations: they alone took 100 s of this test on the GPU box)
    dims = (10, 3, 3) if lx1 == 6 else (6, 2, 2)
    c = mesh3d.box_case_3d(*dims, lx1, lengths=(1.0, 0.6, 0.6), re=400.0, endtime=0.01, ub_func=_ubf, warp=0.0, stretch=stretch)
    c.ub = c.ub * c.mask
    c.spng = np.zeros_like(c.x)
    o = _oracle(c, solvers=False)
    rng = np.random.default_rng(5)
    r = rng.standard_normal((3,) + c.x.shape)
    h2 = (11.0 / 6.0) / o.dt
    ref = np.stack([o.helm_solve(r[k], o.nu, h2) for k in range(3)])
    x, y, z = c.x, c.y, c.z
    q = [np.sin(3.0 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel,) + (c.lx1 - 2,) * 3)]
    its, maps = {}, {}
    for fdm in (0, 1, -1):
        if fdm >= 0:
            monkeypatch.setenv("NSK_HELM_FDM", str(fdm))
        else:
            monkeypatch.delenv("NSK_HELM_FDM")
        h = _hip(c, max_helm_iter=400)
        try:
            out, it1 = h.t_op3(5, r, 3)
            assert _rel(out, ref) < 1e-8, fdm
            a, b = h.alloc(2)
            h.upload3(a, *q)
            h.set_nsteps(4)
            h.matvec(b, a, 0)
            maps[fdm] = h.download3(b)
            st = h.stats()
            assert st["unconverged"] == 0
            its[fdm] = (it1, st["helm_iters"] / 4.0)
        finally:
            h.close()
    print("lx1 %d: Helmholtz iterations (random rhs, per step of the map): Jacobi %s, element-block FDM %s" % (lx1, its[0], its[1]))
    sc = max(np.abs(maps[0][k]).max() for k in range(3))
    for k in range(3):
        assert np.abs(maps[1][k] - maps[0][k]).max() < 1e-7 * sc
    assert its[-1] == its[0] and all(np.array_equal(m0, m1) for m0, m1 in zip(maps[-1], maps[0]))      # default = Jacobi
