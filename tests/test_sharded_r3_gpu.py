"""Element sharding, round 3: what the sharded path was refusing or doing differently from the pinned single-rank operator
(missing 1-3): singular pressure operators (adjoint runs, closed domains: `ortho` over all ranks), the pressure
projection space in shards, the nonlinear map / set_baseflow / time-periodic base flows on shards, composed maps, one packed
halo message per peer, and the chunked all-reduce of nsk_orth on an RCCL communicator.  Virtual ranks on one GPU; the same
protocol across processes is tests/test_multiprocess_gpu.py."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _rel(w, got, ref, ncomp=2):
    num = np.sqrt(sum(np.sum(w * (a - b) ** 2) for a, b in zip(got[:ncomp], ref[:ncomp])))
    den = np.sqrt(sum(np.sum(w * b ** 2) for b in ref[:ncomp]))
    return num / den


def _seedvec(case):
    from nekstab_amd import seed
    qx, qy = seed.add_noise(case)
    return qx, qy


@pytest.mark.parametrize("nranks", [2, 3])
def test_projection_space_in_shards_equals_single_rank(case6, oracle6_nosolve, modes, nranks):
    """residualProj = yes (1cyl.par:30): the sharded solves project onto the previous pressure solutions like the single-rank
    ones (same space size, dots summed over ranks), across TWO consecutive maps -- the space survives a map."""
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    h = NekStabHip(case6, case6.meta["vert"], case6.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8,
                   max_helm_iter=120, max_pres_iter=48)
    try:
        u = modes["dRe_u"].astype(np.float64)
        q = (u[0], u[1], oracle6_nosolve.J12 @ modes["dRe_p"].astype(np.float64) @ oracle6_nosolve.J12.T)
        h.set_nsteps(14)
        vq, vf = h.alloc(2)
        g = ShardGroup(h, case6, nranks)
        g.set_nsteps(14)
        sq, sf = g.alloc(2)
        h.upload(vq, *q); g.upload(sq, *q)
        its = []
        for rep in range(2):
            h.matvec(vf, vq, 0); g.matvec(sf, sq, 0)
            ref, got = h.download(vf), g.download(sf)
            r = _rel(oracle6_nosolve.bm1, got, ref)
            its.append((h.stats()["pres_iters"], g.stats()["pres_iters"]))
            print("map", rep, "nranks", nranks, "rel diff", r, "pressure iterations single / sharded", its[-1])
            assert r < 1e-8
            h.copy(vq, vf); g.copy(sq, sf)
        # the projection space is doing its work in the shards too: same iteration counts as the single-rank run (+-10 %)
        for a, b in its:
            assert abs(a - b) <= max(3, 0.1 * a), its
        g.free([sq, sf]); g.close()
    finally:
        h.close()


@pytest.mark.parametrize("nranks", [2, 3])
def test_singular_pressure_adjoint_on_shards(nranks):
    """Adjoint cylinder (1cyl.usr:126-132 turns 'O' into 'v': E has the constant null space), lx1 = 8: `ortho` takes the mean
    over the Gauss nodes of ALL ranks; sharded adjoint map = single-rank map at solver tolerance."""
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8, adjoint=True)
    assert not case.has_outflow
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8,
                   max_helm_iter=150, max_pres_iter=48)
    try:
        qx, qy = _seedvec(case)
        h.set_nsteps(6)
        vq, vf = h.alloc(2)
        h.upload(vq, qx, qy, np.zeros(h.npres))
        h.matvec(vf, vq, 1)
        ref = h.download(vf)
        g = ShardGroup(h, case, nranks)
        g.set_nsteps(6)
        sq, sf = g.alloc(2)
        g.upload(sq, qx, qy, np.zeros(h.npres))
        g.matvec(sf, sq, 1)
        got = g.download(sf)
        w = np.ones_like(case.x)
        r = _rel(w, got, ref)
        print("adjoint, nranks", nranks, "rel diff", r, "mean pressure", got[2].mean(), ref[2].mean())
        assert r < 1e-8
        assert g.stats()["unconverged"] == 0
        # composed map (transient growth, core/matvec.f:343-346) on shards
        h.matvec(vf, vq, 2); g.matvec(sf, sq, 2)
        assert _rel(w, g.download(sf), h.download(vf)) < 1e-7
        g.free([sq, sf]); g.close()
    finally:
        h.close()


def _cavity(re=3600.0):
    from nekstab_amd import mesh, nekio
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    z = np.load(os.path.join(GOLDEN, "cavity_case.npz"))
    bcs = [(int(a), int(b), np.zeros(5), str(cd)) for (a, b), cd in zip(z["bc_ef"], z["bc_code"])]
    m = nekio.Re2Mesh(2, z["xc"].shape[0], z["xc"], z["yc"], None, [], bcs)
    c2 = mesh.build_case_2d(m, z["vlex"].astype(np.int64), z["bf_u"].astype(np.float64), 6, re=re, endtime=1.0, spng_str=0.0)
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    return c2, J @ z["bf_p"].astype(np.float64) @ J.T


def test_nonlinear_map_and_set_baseflow_on_shards():
    """Closed lid-driven cavity (pressure null space, the geometry family of BASELINE config 5): the nonlinear map
    (nonlinear_forward_map, core/newton_krylov.f:336-378) and a re-linearisation (dt / nsteps from the CFL maximum over all
    ranks) on 3 shards = single rank."""
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    c2, p2 = _cavity()
    h = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], tol_helm=1e-12, tol_pres=1e-5, tol_relative=1, nproj=0,
                   max_helm_iter=150, max_pres_iter=48)
    try:
        g = ShardGroup(h, c2, 3)
        rng = np.random.default_rng(5)
        bump = 1e-2 * np.sin(np.pi * c2.x) * np.sin(np.pi * c2.y / 1.2) * c2.mask
        q = (c2.ub[0] + bump, c2.ub[1] - 0.5 * bump, p2)
        vq, vf = h.alloc(2); sq, sf = g.alloc(2)
        h.upload(vq, *q); g.upload(sq, *q)
        h.set_nsteps(12); g.set_nsteps(12)
        h.nonlinear_map(vf, vq); g.nonlinear_map(sf, sq)
        ref, got = h.download(vf), g.download(sf)
        w = np.ones_like(c2.x)
        r = _rel(w, got, ref)
        print("nonlinear map, 3 shards: rel diff", r)
        assert r < 1e-9
        # new linearisation point = the perturbed state: same dt rule on shards
        h.set_baseflow(vf); g.set_baseflow(sf)
        assert g.nsteps == h.nsteps and abs(g.dt - h.dt) < 1e-15
        h.set_nsteps(8); g.set_nsteps(8)
        pert = (bump, 2.0 * bump, np.zeros(h.npres))
        h.upload(vq, *pert); g.upload(sq, *pert)
        h.matvec(vf, vq, 0); g.matvec(sf, sq, 0)
        r = _rel(w, g.download(sf), h.download(vf))
        print("linearised map about the new base flow, 3 shards: rel diff", r)
        assert r < 1e-8
        # Newton's right-hand side form
        h.nonlinear_map(vf, vq, subtract_q=True); g.nonlinear_map(sf, sq, subtract_q=True)
        assert _rel(w, g.download(sf), h.download(vf)) < 1e-8
        g.free([sq, sf]); g.close()
    finally:
        h.close()


def test_time_periodic_base_flow_on_shards(case6):
    """Floquet (uparam(1) = 3.11): the orbit is integrated by the sharded nonlinear stepper and stored per rank; a linearised
    map over it = the single-rank one.  Short 'period' (T = 0.3) to keep the test cheap."""
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    from nekstab_amd.sharded import ShardGroup
    z = np.load(os.path.join(GOLDEN, "cylinder_upo.npz"))
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, endtime=0.3)
    case.ub[:] = z["u"]
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8,
                   max_helm_iter=150, max_pres_iter=48)
    try:
        J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
        q0 = (z["u"][0], z["u"][1], J @ z["p"] @ J.T)
        g = ShardGroup(h, case, 2)
        a0, ae, vq, vf = h.alloc(4)
        b0, be, sq, sf = g.alloc(4)
        h.upload(a0, *q0); g.upload(b0, *q0)
        h.set_orbit(a0, spng_str=1.7, end=ae); g.set_orbit(b0, spng_str=1.7, end=be)
        assert g.nsteps == h.nsteps
        w = np.ones_like(case.x)
        r = _rel(w, g.download(be), h.download(ae))
        print("orbit end state, 2 shards: rel diff", r)
        assert r < 1e-9
        qx, qy = _seedvec(case)
        h.upload(vq, qx, qy, np.zeros(h.npres)); g.upload(sq, qx, qy, np.zeros(h.npres))
        h.matvec(vf, vq, 0); g.matvec(sf, sq, 0)
        r = _rel(w, g.download(sf), h.download(vf))
        print("linearised map over the stored orbit, 2 shards: rel diff", r)
        assert r < 1e-8
        g.free([b0, be, sq, sf]); g.close()
    finally:
        h.close()


def test_time_periodic_base_flow_on_hexahedral_shards():
    """The same on hexahedra (round 4: `nsk_set_orbit` / `nsk_group_set_orbit` store the twelve dealiasing-mesh constants per step):
    extruded cylinder, the shedding state as the initial base flow, 2 and 3 shards against the single rank -- orbit end state, then a
    direct and an adjoint map over the stored orbit."""
    from nekstab_amd import mesh, mesh3d, seed
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    from nekstab_amd.sharded import ShardGroup
    z = np.load(os.path.join(GOLDEN, "cylinder_upo.npz"))
    c2 = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, endtime=0.15)
    c2.ub[:] = z["u"]
    c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, nproj=0,
                   max_helm_iter=400, max_pres_iter=192)
    try:
        J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
        q0 = (mesh3d.extrude_field(z["u"][0], 2), mesh3d.extrude_field(z["u"][1], 2), np.zeros(c3.x.shape), mesh3d.extrude_pressure(J @ z["p"] @ J.T, 2))
        a0, ae, vq, vf = h.alloc(4)
        h.upload3(a0, *q0)
        h.set_orbit(a0, spng_str=1.7, end=ae)
        ref_end = h.download3(ae)
        qx, qy = seed.add_noise(c2)
        q = (mesh3d.extrude_field(qx, 2) * np.cos(2 * np.pi * c3.z), mesh3d.extrude_field(qy, 2), mesh3d.extrude_field(qx, 2) * np.sin(2 * np.pi * c3.z) * c3.mask,
             np.zeros(h.npres))
        h.upload3(vq, *q)
        refs = []
        for mode in (0, 1):
            h.matvec(vf, vq, mode); refs.append(h.download3(vf))
        for nranks in (2, 3):
            g = ShardGroup(h, c3, nranks)
            b0, be, sq, sf = g.alloc(4)
            g.upload3(b0, *q0)
            g.set_orbit(b0, spng_str=1.7, end=be)
            assert g.nsteps == h.nsteps
            got = g.download3(be)
            sc = max(np.abs(ref_end[k]).max() for k in range(2))
            err = max(np.abs(got[k] - ref_end[k]).max() for k in range(3)) / sc
            print("hexahedral orbit end state,", nranks, "shards: max diff", err)
            assert err < 1e-8
            g.upload3(sq, *q)
            for mode in (0, 1):
                g.matvec(sf, sq, mode)
                got = g.download3(sf)
                sc = max(np.abs(refs[mode][k]).max() for k in range(3))
                err = max(np.abs(got[k] - refs[mode][k]).max() for k in range(3)) / sc
                print("mode", mode, "map over the stored orbit,", nranks, "hexahedral shards: max diff", err)
                assert err < 1e-6
            g.free([b0, be, sq, sf]); g.close()
    finally:
        h.close()


def test_backstep_adjoint_hexahedra_chebyshev_coarse_on_shards(monkeypatch):
    """BASELINE config 4's case in small: backward-facing step extruded in z, ADJOINT map (all-Dirichlet: singular pressure),
    hexahedral kernels, sparse coarse operator with the Chebyshev polynomial forced (what 5e4 vertices need), 2 and 3 shards."""
    from nekstab_amd import mesh, mesh3d
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    monkeypatch.setenv("NSK_COARSE_ITER", "1")
    c2 = mesh.load_case_npz(os.path.join(GOLDEN, "backstep_case.npz"), 6, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
    c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-11, tol_pres=1e-3, tol_relative=1, max_helm_iter=200, max_pres_iter=192)
    try:
        assert not c3.has_outflow
        x, y, zc = c3.x, c3.y, c3.z
        q = [np.sin(1.3 * x + zc) * np.cos(2.0 * y) * c3.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - zc) * c3.mask,
             np.sin(x + y) * np.cos(2.0 * np.pi * zc) * c3.mask, np.zeros(h.npres)]
        h.set_nsteps(3)
        vq, vf = h.alloc(2)
        h.upload3(vq, *q)
        h.matvec(vf, vq, 1)
        ref = h.download3(vf)
        sc = max(np.abs(ref[k]).max() for k in range(3))
        for nranks in (2, 3):
            g = ShardGroup(h, c3, nranks)
            g.set_nsteps(3)
            sq, sf = g.alloc(2)
            g.upload3(sq, *q)
            g.matvec(sf, sq, 1)
            got = g.download3(sf)
            err = max(np.abs(got[k] - ref[k]).max() for k in range(3)) / sc
            print("extruded back-step, adjoint, Chebyshev coarse solve,", nranks, "shards: max diff", err)
            assert err < 1e-6
            assert g.stats()["unconverged"] == 0
            g.free([sq, sf]); g.close()
        # restarted GMRES on shards (config 5 needs ~50 iterations per solve: more than one cycle): cycles of 20 on both sides
        h.set_option("gmres_cycle", 20)
        h.matvec(vf, vq, 1)
        ref = h.download3(vf)
        its = h.stats()["max_pres_iter"]
        g = ShardGroup(h, c3, 2)
        g.set_nsteps(3)
        sq, sf = g.alloc(2)
        g.upload3(sq, *q)
        g.matvec(sf, sq, 1)
        got = g.download3(sf)
        err = max(np.abs(got[k] - ref[k]).max() for k in range(3)) / sc
        print("restarted GMRES (cycles of 20), 2 shards: max diff", err, "worst solve", its, "/", g.stats()["max_pres_iter"], "iterations")
        assert its > 20 and err < 1e-6
        g.free([sq, sf]); g.close()
    finally:
        h.close()


def test_orth_chunked_allreduce_on_rccl_communicator(hip6, case6):
    """nsk_orth on an RCCL rank: the coefficient all-reduces run in chunks on a second stream while the next chunk's dots are
    computed.  On the one-GPU box the communicator has one rank (RCCL refuses two ranks per device), which exercises the
    streams, events and call order; the result must equal the single all-reduce form bit for bit."""
    from nekstab_amd.sharded import ShardRank
    uid = ShardRank.new_unique_id(hip6.lib)
    s = ShardRank(hip6, case6, 0, 1, uid)
    try:
        rng = np.random.default_rng(7)
        nq = 40
        vecs = s.alloc(nq + 2)
        shape_v, shape_p = case6.x.shape, (case6.nel, 4, 4)
        for v in vecs:
            s.upload(v, rng.standard_normal(shape_v), rng.standard_normal(shape_v), rng.standard_normal(shape_p))
        Q = vecs[:nq]
        # orthonormalise Q with the plain path first
        s.set_option("orth_overlap", 0)
        for j in range(nq):
            s.orth(Q[j], Q[:j])
        f0, f1 = vecs[nq], vecs[nq + 1]
        s.copy(f1, f0)
        h0, b0 = s.orth(f0, Q)
        s.set_option("orth_overlap", 1)
        h1, b1 = s.orth(f1, Q)
        assert np.array_equal(h0, h1) and b0 == b1
        a, b = s.download_local(f0), s.download_local(f1)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        s.free(vecs)
    finally:
        s.close()


def test_sharded_adjoint_spectrum_against_reference_table(spectre):
    """End to end on shards at the production settings: k_dim = 120 adjoint Arnoldi at lx1 = 8 on two element shards; the
    leading pair and the first converged wake rows of the reference's Spectre_Ha.dat within the stated 5e-6."""
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.settings import production_context
    from nekstab_amd.sharded import ShardGroup
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8, adjoint=True)
    h = production_context(case)
    try:
        g = ShardGroup(h, case, 2)
        qx, qy = seed.add_noise(case)
        v0, v1 = g.alloc(2)
        g.upload(v0, qx, qy, np.zeros(h.npres))
        g.scal(v0, 1.0 / g.norm(v0))
        g.matvec(v1, v0, 1)
        res = krylov.krylov_schur(g, v1, 120, mode=1, schur_tgt=0)
        tab = spectre["Ha"]
        n = 0
        for r in tab:
            if r[2] >= 1e-8 or r[1] < 0:
                continue
            zr = complex(r[0], r[1])
            j = int(np.argmin(np.abs(res.vals - zr)))
            if res.residual[j] > 1e-8:
                continue
            d = abs(res.vals[j] - zr)
            print("Ha %.7f%+.7fi  sharded %.9f%+.9fi (res %.0e)  diff %.1e" % (zr.real, zr.imag, res.vals[j].real, res.vals[j].imag, res.residual[j], d))
            assert d < 5e-6
            n += 1
        assert n >= 1
        g.close()
    finally:
        h.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_host_checked_convergence_equals_budgeted_launches(case6, oracle6_nosolve, modes, dim):
    """Option shard_hostcheck (the default once a transport is attached: a launched iteration costs its halo exchange and its
    all-reduce whether the solve has converged or not): the host reads the device's convergence flags and stops issuing
    iterations.  Launches that find their solve converged change nothing, so the maps are BIT-identical to the budgeted ones
    (eager and captured), with the same iteration counts."""
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    if dim == 2:
        case = case6
        u = modes["dRe_u"].astype(np.float64)
        q = (u[0], u[1], oracle6_nosolve.J12 @ modes["dRe_p"].astype(np.float64) @ oracle6_nosolve.J12.T)
        kw = dict(tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8, max_helm_iter=120, max_pres_iter=48)
        nst, R = 14, 3
    else:
        from nekstab_amd import mesh3d
        ubf = lambda x, y, z: np.stack([1.0 - 0.3 * y * y + 0.1 * np.sin(x + z), 0.2 * np.cos(x) * y + 0.1 * z, 0.15 * np.sin(y + 0.5 * z)])
        case = mesh3d.box_case_3d(4, 3, 2, 6, lengths=(2.0, 1.0, 0.8), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=ubf, warp=0.05)
        x, y, z = case.x, case.y, case.z
        q = (np.sin(1.3 * x + z) * np.cos(2.0 * y) * case.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * case.mask,
             np.sin(x + y) * np.cos(2.0 * z) * case.mask, np.zeros((case.nel, 4, 4, 4)))
        kw = dict(tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, max_helm_iter=200, max_pres_iter=48)
        nst, R = 4, 2
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], **kw)
    try:
        out, its = {}, {}
        for name, opts in (("graph", {}), ("eager", {"shard_graph": 0}), ("hostcheck", {"shard_hostcheck": 1})):
            g = ShardGroup(h, case, R)
            for k, v in opts.items():
                g.set_option(k, v)
            g.set_nsteps(nst)
            a, b = g.alloc(2)
            (g.upload if dim == 2 else g.upload3)(a, *q)
            res = []
            for rep in range(2):
                g.matvec(b, a, 0)
                res.append((g.download if dim == 2 else g.download3)(b))
                g.copy(a, b)
            out[name] = res
            st = g.stats()
            its[name] = (st["helm_iters"], st["pres_iters"])
            g.free([a, b]); g.close()
        print("iterations of the last map (velocity, pressure):", its)
        for name in ("eager", "hostcheck"):
            for rep in range(2):
                for x0, x1 in zip(out["graph"][rep], out[name][rep]):
                    assert np.array_equal(x0, x1), (name, rep)
            assert its[name] == its["graph"]
    finally:
        h.close()


def test_halo_interior_overlap_equals_serial_order(case6, oracle6_nosolve, modes):
    """Option halo_overlap (north_star: "halo overlapped with interior work"): the velocity solve on shards launches the
    workgroups of the BOUNDARY elements first (a shard keeps them at the front: nsk_shard_elems), sends their halo on a second
    stream while the interior workgroups run, and lets the all-reduce wait for both.  Same kernels on the same data in an order
    the events make equivalent: bit-identical maps, with budgeted launches and with host-checked convergence; and equal to the
    single-rank map to the solver tolerance."""
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    u = modes["dRe_u"].astype(np.float64)
    q = (u[0], u[1], oracle6_nosolve.J12 @ modes["dRe_p"].astype(np.float64) @ oracle6_nosolve.J12.T)
    h = NekStabHip(case6, case6.meta["vert"], case6.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8, max_helm_iter=120, max_pres_iter=48)
    try:
        h.set_nsteps(10)
        vq, vf = h.alloc(2)
        h.upload(vq, *q); h.matvec(vf, vq, 0)
        ref = h.download(vf)
        out = {}
        for name, opts in (("serial", {"shard_graph": 0}), ("overlap", {"shard_graph": 0, "halo_overlap": 1}),
                           ("overlap + host check", {"shard_hostcheck": 1, "halo_overlap": 1})):
            g = ShardGroup(h, case6, 3)
            # boundary elements first: every rank's element list starts with the elements that touch another rank
            for r in range(3):
                e = g.elems[r]
                assert sorted(e.tolist()) == np.where(g.part == r)[0].tolist() and not np.all(np.diff(e) > 0)
            for k, v in opts.items():
                g.set_option(k, v)
            g.set_nsteps(10)
            a, b = g.alloc(2)
            g.upload(a, *q)
            g.matvec(b, a, 0)
            out[name] = g.download(b)
            g.free([a, b]); g.close()
        for name in ("overlap", "overlap + host check"):
            for x0, x1 in zip(out["serial"], out[name]):
                assert np.array_equal(x0, x1), name
        assert _rel(oracle6_nosolve.bm1, out["overlap"], ref) < 1e-8
    finally:
        h.close()


def test_halo_interior_overlap_hexahedra():
    """The same on hexahedral shards (three velocity components per halo message, workgroup -> element map with the XCD
    de-interleave inside each of the two launches): bit-identical to the serial order, two ranks and three."""
    from nekstab_amd import mesh3d
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    ubf = lambda x, y, z: np.stack([1.0 - 0.3 * y * y + 0.1 * np.sin(x + z), 0.2 * np.cos(x) * y + 0.1 * z, 0.15 * np.sin(y + 0.5 * z)])
    c = mesh3d.box_case_3d(6, 4, 3, 6, lengths=(3.0, 1.5, 1.0), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=ubf, warp=0.05)
    x, y, z = c.x, c.y, c.z
    q = (np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel, 4, 4, 4)))
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, max_helm_iter=200, max_pres_iter=48)
    try:
        for R in (2, 3):
            out = {}
            for name, opts in (("serial", {"shard_graph": 0}), ("overlap", {"shard_graph": 0, "halo_overlap": 1}), ("overlap + host check", {"shard_hostcheck": 1, "halo_overlap": 1})):
                g = ShardGroup(h, c, R)
                for k, v in opts.items():
                    g.set_option(k, v)
                g.set_nsteps(4)
                a, b = g.alloc(2)
                g.upload3(a, *q)
                g.matvec(b, a, 0)
                out[name] = g.download3(b)
                g.free([a, b]); g.close()
            for name in ("overlap", "overlap + host check"):
                for x0, x1 in zip(out["serial"], out[name]):
                    assert np.array_equal(x0, x1), (R, name)
    finally:
        h.close()


def test_shards_at_lx1_12_one_element_per_workgroup():
    """BASELINE configs[2]'s order (lx1 = 12: one element per workgroup, the totals path of many workgroups) on three shards:
    serial order, halo / interior overlap and host-checked convergence all equal to one another bit for bit and to the single
    rank to the solver tolerance."""
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 12)
    modes = np.load(os.path.join(GOLDEN, "cylinder_modes.npz"))
    u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), 12) * case.mask
    q = (u[0], u[1], np.zeros((case.nel, 10, 10)))
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-5, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=96)
    try:
        h.set_nsteps(4)
        vq, vf = h.alloc(2)
        h.upload(vq, *q); h.matvec(vf, vq, 0)
        ref = h.download(vf)
        out = {}
        for name, opts in (("serial", {"shard_graph": 0}), ("overlap", {"shard_graph": 0, "halo_overlap": 1}), ("host check", {"shard_hostcheck": 1}), ("graph", {})):
            g = ShardGroup(h, case, 3)
            for k, v in opts.items():
                g.set_option(k, v)
            g.set_nsteps(4)
            a, b = g.alloc(2)
            g.upload(a, *q)
            g.matvec(b, a, 0)
            out[name] = g.download(b)
            g.free([a, b]); g.close()
        for name in ("overlap", "host check", "graph"):
            for x0, x1 in zip(out["serial"], out[name]):
                assert np.array_equal(x0, x1), name
        sc = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
        assert max(np.abs(out["serial"][k] - ref[k]).max() for k in range(2)) < 1e-8 * sc
    finally:
        h.close()
