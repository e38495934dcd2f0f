"""Rank-local set-up: every rank builds its context over ITS sub-mesh (own elements + two rings of
neighbours: nsk_init_local), the few global facts are exchanged (volume, CFL maximum, the coarse rows of the owned vertices:
nsk_local_info / nsk_local_rows / nsk_local_finish) and the shard is cut with halos keyed by global ids
(nsk_shard_create_local).  Held here against the shards cut from a whole-mesh parent and against the single-rank operator:
same dt / nsteps, same maps, on quadrilaterals (outflow and singular pressure operators, the projection space) and on
hexahedra (dense, block-circulant and polynomial coarse solves).  Virtual ranks on one GPU; across processes:
tests/test_multiprocess_gpu.py."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

KW = dict(tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, max_helm_iter=200, max_pres_iter=64)


def _rel(w, got, ref, ncomp=2):
    num = np.sqrt(sum(np.sum(w * (a - b) ** 2) for a, b in zip(got[:ncomp], ref[:ncomp])))
    den = np.sqrt(sum(np.sum(w * b ** 2) for b in ref[:ncomp]))
    return num / den


@pytest.mark.parametrize("nranks,mode", [(2, 0), (3, 0), (3, 1)])
def test_local_setup_equals_whole_mesh_setup_2d(case6, oracle6_nosolve, modes, nranks, mode):
    """The cylinder (lx1 = 6, 1996 elements): direct maps (outflow) and adjoint maps (singular pressure operator: `ortho`, the
    shifted coarse operator), projection space on, two consecutive maps."""
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup, local_parents, partition_rcb
    from nekstab_amd import mesh
    case = case6 if mode == 0 else mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6, adjoint=True)      # 'O' -> 'v': singular
    part = partition_rcb(case, nranks)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], nproj=8, **KW)
    P = []
    try:
        u = modes["dRe_u"].astype(np.float64)
        q = (u[0] * case.mask, u[1] * case.mask, oracle6_nosolve.J12 @ modes["dRe_p"].astype(np.float64) @ oracle6_nosolve.J12.T)
        P, _ = local_parents(case, nranks, part, nproj=8, **KW)
        for p in P:
            assert p.nel < 0.8 * case.nel, (p.nel, case.nel)                   # a sub-mesh, not the mesh
            assert p.nsteps == h.nsteps and abs(p.dt - h.dt) < 1e-15
        print("elements per sub-mesh:", [p.nel for p in P], "of", case.nel, "(owned:", [int((part == r).sum()) for r in range(nranks)], ")")
        gw = ShardGroup(h, case, nranks, part)
        gl = ShardGroup(P, case, nranks, part)
        # element operators across shards: dssum and E
        rng = np.random.default_rng(0)
        f = rng.standard_normal(case.x.shape)
        assert np.abs(gl.group_test(0, f) - h.t_dssum(f)).max() < 1e-11
        pp = rng.standard_normal((case.nel, 4, 4))
        er = h.t_eapply(pp)
        assert np.abs(gl.group_test(1, pp) - er).max() < 1e-11 * np.abs(er).max()
        for g in (gw, gl):
            g.set_nsteps(12)
        h.set_nsteps(12)
        vq, vf = h.alloc(2)
        wq, wf = gw.alloc(2)
        lq, lf = gl.alloc(2)
        h.upload(vq, *q); gw.upload(wq, *q); gl.upload(lq, *q)
        for rep in range(2):
            h.matvec(vf, vq, mode); gw.matvec(wf, wq, mode); gl.matvec(lf, lq, mode)
            ref, a, b = h.download(vf), gw.download(wf), gl.download(lf)
            r1, r2 = _rel(oracle6_nosolve.bm1, b, a), _rel(oracle6_nosolve.bm1, b, ref)
            its = (h.stats()["pres_iters"], gw.stats()["pres_iters"], gl.stats()["pres_iters"])
            print("map", rep, "nranks", nranks, "mode", mode, "local vs whole-mesh shards", r1, "vs single rank", r2, "pressure iterations", its)
            assert r1 < 1e-8 and r2 < 1e-8
            assert abs(its[2] - its[1]) <= max(3, 0.1 * its[1]), its
            h.copy(vq, vf); gw.copy(wq, wf); gl.copy(lq, lf)
        gw.close(); gl.close()
    finally:
        for p in P:
            p.close()
        h.close()


def _check_halo_plans(g, c, part, nranks):
    """The exchange plans the LIBRARY derived on every rank (nsk_shard_halo_counts) against the host-side derivation from the
    global numbering (sharded.velocity_halo_plan), and against one another: what r sends to p is what p expects from r."""
    from nekstab_amd.sharded import shard_halo_counts, velocity_halo_plan
    gid = np.asarray(c.gid).reshape(c.nel, -1)
    cnt = [shard_halo_counts(g.lib, g.ctx[r], nranks) for r in range(nranks)]
    for r in range(nranks):
        vel, ps, pr = cnt[r]
        plan = velocity_halo_plan(gid, part, r)
        assert {p: len(v) for p, v in plan.items()} == {p: int(vel[p]) for p in range(nranks) if vel[p]}, r
        for p in range(nranks):
            assert vel[p] == cnt[p][0][r] and ps[p] == cnt[p][2][r] and pr[p] == cnt[p][1][r], (r, p)
        assert vel[r] == 0 and ps[r] == 0 and pr[r] == 0
    print("halo plans of %d ranks agree: velocity messages %s nodes, pressure messages up to %d dofs" %
          (nranks, sorted({int(v) for r in range(nranks) for v in cnt[r][0] if v})[::max(1, nranks // 2)], max(int(cnt[r][1].max()) for r in range(nranks))))


def _maps3(c, nranks, q, nst=4, tol=1e-8, **kw):
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup, local_parents, partition_rcb
    part = partition_rcb(c, nranks)
    check_plan = kw.pop("_check_plan", False)
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], **dict(KW, **kw))
    if check_plan:
        kw = dict(kw)
    P = []
    try:
        ns0 = h.nsteps
        h.set_nsteps(nst)
        vq, vf = h.alloc(2)
        h.upload3(vq, *q)
        h.matvec(vf, vq, 0)
        ref = h.download3(vf)
        P, _ = local_parents(c, nranks, part, **dict(KW, **kw))
        for p in P:
            assert p.nsteps == ns0 and abs(p.dt - h.dt) < 1e-15
        g = ShardGroup(P, c, nranks, part)
        if check_plan:
            _check_halo_plans(g, c, part, nranks)
        g.set_nsteps(nst)
        sq, sf = g.alloc(2)
        g.upload3(sq, *q)
        g.matvec(sf, sq, 0)
        got = g.download3(sf)
        sc = max(np.abs(ref[k]).max() for k in range(3))
        err = max(np.abs(got[k] - ref[k]).max() for k in range(3)) / sc
        its = (h.stats()["pres_iters"], g.stats()["pres_iters"])
        print("elements per sub-mesh:", [p.nel for p in P], "of", c.nel, "velocity difference", err, "pressure iterations", its)
        assert err < tol
        assert abs(its[0] - its[1]) <= max(3, 0.15 * its[0]), its
        g.close()
    finally:
        for p in P:
            p.close()
        h.close()


@pytest.mark.parametrize("nranks", [2, 3])
def test_local_setup_equals_single_rank_3d_box(nranks):
    """Hexahedra, warped box with an outflow (dense coarse inverse): fast-diagonalisation Schwarz factors, patch gather tables
    and coarse rows from the sub-mesh."""
    from nekstab_amd import mesh3d
    ubf = lambda x, y, z: np.stack([1.0 - 0.3 * y * y + 0.1 * np.sin(x + z), 0.2 * np.cos(x) * y + 0.1 * z, 0.15 * np.sin(y + 0.5 * z)])
    c = mesh3d.box_case_3d(6, 4, 3, 6, lengths=(3.0, 1.5, 1.0), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=ubf, warp=0.05)
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel, 4, 4, 4))]
    _maps3(c, nranks, q)


@pytest.mark.parametrize("coarse", ["circulant", "chebyshev"])
def test_local_setup_3d_extruded_cylinder(monkeypatch, coarse):
    """The cylinder extruded over 4 periodic layers (config 4's family): the coarse rows gathered from three ranks must carry
    the block-circulant structure the exact coarse solve detects (and feed the Chebyshev polynomial the same operator)."""
    from nekstab_amd import mesh, mesh3d
    monkeypatch.setenv("NSK_COARSE_ITER", "1" if coarse == "chebyshev" else "0")
    monkeypatch.setenv("NSK_COARSE_CIRC", "1" if coarse == "circulant" else "0")
    c2 = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    c3 = mesh3d.extrude_case(c2, 4, 2.0, periodic=True)
    x, y, z = c3.x, c3.y, c3.z
    env = np.exp(-0.05 * (x - 2.0) ** 2 - 0.2 * y ** 2) * c3.mask
    q = [env * np.sin(y + np.pi * z), env * np.cos(x) * np.cos(np.pi * z), env * np.sin(x + y) * np.sin(np.pi * z), np.zeros((c3.nel, 4, 4, 4))]
    _maps3(c3, 3, q, nst=3, tol=1e-7)


def test_local_setup_3d_closed_cavity_singular_operator(monkeypatch):
    """The lid-driven cavity extruded over 5 periodic layers (config 5's kind: closed, singular pressure operator): `ortho` over
    the ranks' Gauss nodes with the gathered count, the shifted wavenumber-0 block of the block-circulant coarse solve built from
    gathered rows, on two rank-local shards."""
    from nekstab_amd import mesh3d
    from tests.test_3d_gpu import _cavity_2d
    monkeypatch.setenv("NSK_COARSE_ITER", "0")
    monkeypatch.setenv("NSK_COARSE_CIRC", "1")
    c2, _ = _cavity_2d(3600.0)
    c3 = mesh3d.extrude_case(c2, 5, 1.0, periodic=True)
    assert not c3.has_outflow
    x, y, z = c3.x, c3.y, c3.z
    q = [np.sin(np.pi * x) * np.cos(2 * np.pi * z) * c3.mask, np.sin(2 * np.pi * y / 1.2) * np.sin(2 * np.pi * z) * c3.mask,
         np.sin(np.pi * x) * np.sin(np.pi * y / 1.2) * c3.mask, np.zeros((c3.nel, 4, 4, 4))]
    _maps3(c3, 2, q, nst=3, tol=1e-7, nproj=4)


def test_eight_rank_local_shards_closed_hexahedral_box():
    """Eight ranks before the node arrives: a closed hexahedral box (singular pressure operator, `ortho`
    over all ranks) on EIGHT rank-local shards (16 elements each) with the pressure projection space: the exchange plans of
    the eight ranks -- as the library built them from each rank's sub-mesh -- agree with one another and with the host-side
    derivation, and the sharded map equals the single-rank one at solver tolerance."""
    from nekstab_amd import mesh3d
    ubf = lambda x, y, z: np.stack([np.sin(np.pi * x / 4.0) * np.cos(np.pi * y / 2.0) * 0.5 + 0.2, -np.cos(np.pi * x / 4.0) * np.sin(np.pi * y / 2.0) * 0.5, 0.1 * np.sin(np.pi * z / 2.0) + 0.0 * x])
    c = mesh3d.box_case_3d(8, 4, 4, 6, lengths=(4.0, 2.0, 2.0), re=40.0, endtime=0.05, ub_func=ubf, warp=0.04)        # all walls: no outflow
    c.ub = c.ub * c.mask
    assert not c.has_outflow and c.nel == 128
    x, y, z = c.x, c.y, c.z
    q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
         np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel, 4, 4, 4))]
    _maps3(c, 8, q, nst=4, tol=1e-7, nproj=4, _check_plan=True)
