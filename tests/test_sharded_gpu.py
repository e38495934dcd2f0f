"""Element sharding with virtual ranks on one GPU: the sharded time-stepper (dssum halo,
Schwarz-overlap halo, all-reduced dot products and coarse restriction) reproduces the single-rank
result, and the host Arnoldi runs unchanged on sharded vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_shared_context(request):
    """These tests change the tolerances of the session-wide context: put the fixture's own back (the kernel tests rely on them
    whatever the order the files run in)."""
    yield
    if "hip6" in request.fixturenames:
        request.getfixturevalue("hip6").set_tolerances(1e-13, 1e-13, 0)


def _mode(o, modes):
    u = modes["dRe_u"].astype(np.float64)
    return u[0], u[1], o.J12 @ modes["dRe_p"].astype(np.float64) @ o.J12.T


@pytest.mark.parametrize("nranks", [2, 4])
def test_sharded_dssum_and_eapply(hip6, case6, oracle6_nosolve, nranks):
    from nekstab_amd.sharded import ShardGroup
    o = oracle6_nosolve
    rng = np.random.default_rng(3)
    u = rng.standard_normal(case6.x.shape)
    p = rng.standard_normal((case6.nel, 4, 4))
    g = ShardGroup(hip6, case6, nranks)
    got = g.group_test(0, u)
    assert np.abs(got - o.dssum(u)).max() < 1e-12 * np.abs(u).max() * 8
    ref = hip6.t_eapply(p)
    got = g.group_test(1, p)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    g.close()


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_sharded_matvec_equals_single_rank(hip6, case6, oracle6_nosolve, modes, nranks):
    from nekstab_amd.sharded import ShardGroup, partition_rcb
    part = partition_rcb(case6, nranks)
    assert sorted(np.unique(part)) == list(range(nranks))
    assert np.bincount(part).min() >= case6.nel // nranks - 1              # balanced
    q = _mode(oracle6_nosolve, modes)
    hip6.set_tolerances(1e-12, 1e-6, 1)
    hip6.set_nsteps(4)
    vq, vf = hip6.alloc(2)
    hip6.upload(vq, *q)
    hip6.matvec(vf, vq, 0)
    ref = hip6.download(vf)
    g = ShardGroup(hip6, case6, nranks, part)
    g.set_nsteps(4)
    sq, sf = g.alloc(2)
    g.upload(sq, *q)
    assert abs(g.dot(sq, sq) - hip6.dot(vq, vq)) < 1e-12 * hip6.dot(vq, vq)
    g.matvec(sf, sq, 0)
    got = g.download(sf)
    w = oracle6_nosolve.bm1
    num = np.sqrt(sum(np.sum(w * (a - b) ** 2) for a, b in zip(got[:2], ref[:2])))
    den = np.sqrt(sum(np.sum(w * b ** 2) for b in ref[:2]))
    print("nranks", nranks, "rel diff vs single rank", num / den)
    assert num / den < 1e-9                     # iterative solves: equal to the solver tolerance
    assert np.abs(got[2] - ref[2]).max() < 1e-5 * np.abs(ref[2]).max()
    g.free([sq, sf]); g.close()
    hip6.free([vq, vf]); hip6.set_nsteps(100)


def test_sharded_arnoldi_matches_single_rank(hip6, case6):
    """The host Arnoldi (nekstab_amd/krylov.py) runs unchanged on sharded vectors: same Hessenberg
    matrix as the single-rank run (globally summed projections, rank-local updates)."""
    from nekstab_amd import krylov, seed
    from nekstab_amd.sharded import ShardGroup
    hip6.set_tolerances(1e-11, 1e-2, 1)
    hip6.set_nsteps(20)
    qx, qy = seed.add_noise(case6)
    pr = np.zeros((case6.nel, 4, 4))
    v0 = hip6.alloc(1)[0]
    hip6.upload(v0, qx, qy, pr)
    r1 = krylov.krylov_schur(hip6, v0, 6, schur_tgt=0)
    g = ShardGroup(hip6, case6, 3)
    g.set_nsteps(20)
    s0 = g.alloc(1)[0]
    g.upload(s0, qx, qy, pr)
    r2 = krylov.krylov_schur(g, s0, 6, schur_tgt=0)
    assert np.abs(r1.H - r2.H).max() < 1e-9 * np.abs(r1.H).max()
    assert np.abs(np.sort_complex(r1.vals) - np.sort_complex(r2.vals)).max() < 1e-9
    g.free(r2.Q + [s0]); g.close()
    hip6.free(r1.Q + [v0]); hip6.set_nsteps(100)


def test_rccl_transport_single_rank_plumbing(hip6, case6, oracle6_nosolve, modes):
    """The RCCL path (dlopen, communicator, all-reduce on the library stream) on a 1-rank
    communicator: same result as the plain context.  (send/recv between ranks needs >= 2 GPUs.)"""
    from nekstab_amd.sharded import ShardRank
    q = _mode(oracle6_nosolve, modes)
    hip6.set_tolerances(1e-12, 1e-6, 1)
    hip6.set_nsteps(3)
    vq, vf = hip6.alloc(2)
    hip6.upload(vq, *q)
    hip6.matvec(vf, vq, 0)
    ref = hip6.download(vf)
    uid = ShardRank.new_unique_id(hip6.lib)
    assert len(uid) == 128
    s = ShardRank(hip6, case6, 0, 1, uid)
    s.set_nsteps(3)
    a, b = s.alloc(2)
    s.upload(a, *q)
    assert abs(s.dot(a, a) - hip6.dot(vq, vq)) < 1e-12 * hip6.dot(vq, vq)
    s.matvec(b, a, 0)
    got = s.download_local(b)
    for x, y in zip(got, ref):
        assert np.abs(x - y).max() <= 1e-12 * np.abs(y).max()
    # the mode bench.py --gpus N tries first on a node: the sharded step as one captured graph per step class WITH the RCCL
    # calls inside (here: the all-reduces of a one-rank communicator -- what one GPU can exercise of it); bit-identical to the
    # eager, host-checked run above
    s.set_option("shard_graph", 1)
    s.set_option("shard_hostcheck", 0)
    s.matvec(b, a, 0)
    for x, y in zip(s.download_local(b), got):
        assert np.array_equal(x, y)
    assert s.stats()["recaptures"] > 0
    s.free([a, b]); s.close()
    hip6.free([vq, vf]); hip6.set_nsteps(100)


@pytest.mark.parametrize("nranks,lx1", [(2, 6), (3, 6), (2, 10)])
def test_sharded_hexahedral_matvec_equals_single_rank(nranks, lx1):
    """Element sharding of a hexahedral context (BASELINE configs 4 and 5 are 3-D on 8 GPUs): dssum halo of three
    velocity components, halo of the Schwarz patch layers (face, edge and corner neighbours on other ranks),
    all-reduced totals of both Gram-Schmidt passes and of the coarse restriction; virtual ranks on one GPU."""
    from nekstab_amd import mesh3d
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup, partition_rcb
    ubf = lambda x, y, z: np.stack([1.0 - 0.3 * y * y + 0.1 * np.sin(x + z), 0.2 * np.cos(x) * y + 0.1 * z, 0.15 * np.sin(y + 0.5 * z)])
    # (lx1 = 10: the sharded step launches k_convect_mfma<10>, k_schwarz_q<10> and k_divgs_c3<10> as the single-rank context does)
    c = mesh3d.box_case_3d(4, 3, 2, lx1, lengths=(2.0, 1.0, 0.8), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=ubf, warp=0.05)
    m = lx1 - 2
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, max_helm_iter=200,
                   max_pres_iter=48)
    try:
        x, y, z = c.x, c.y, c.z
        q = [np.sin(1.3 * x + z) * np.cos(2.0 * y) * c.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * c.mask,
             np.sin(x + y) * np.cos(2.0 * z) * c.mask, np.zeros((c.nel, m, m, m))]
        h.set_nsteps(4)
        vq, vf = h.alloc(2)
        h.upload3(vq, *q)
        h.matvec(vf, vq, 0)
        ref = h.download3(vf)
        part = partition_rcb(c, nranks)
        assert sorted(np.unique(part)) == list(range(nranks))
        g = ShardGroup(h, c, nranks, part)
        rng = np.random.default_rng(0)
        u = rng.standard_normal(c.x.shape)
        assert np.abs(g.group_test(0, u) - h.t_dssum(u)).max() < 1e-12 * 8
        p = rng.standard_normal((c.nel, m, m, m))
        er = h.t_eapply(p)
        assert np.abs(g.group_test(1, p) - er).max() < 1e-12 * np.abs(er).max()
        g.set_nsteps(4)
        sq, sf = g.alloc(2)
        g.upload3(sq, *q)
        assert abs(g.dot(sq, sq) - h.dot(vq, vq)) < 1e-12 * h.dot(vq, vq)
        g.matvec(sf, sq, 0)
        got = g.download3(sf)
        sc = max(np.abs(ref[k]).max() for k in range(3))
        for k in range(3):
            assert np.abs(got[k] - ref[k]).max() < 1e-8 * sc
        assert np.abs(got[3] - ref[3]).max() < 1e-4 * np.abs(ref[3]).max()
        g.free([sq, sf]); g.close()
    finally:
        h.close()


def test_sharded_step_graph_equals_eager(hip6, case6, oracle6_nosolve, modes):
    """The sharded time step captured in one hipGraph per step class (all ranks of the process, their halo copies and
    reductions) gives bit for bit what the eager launches give, map after map while the launch budgets adapt."""
    import time
    from nekstab_amd.sharded import ShardGroup
    q = _mode(oracle6_nosolve, modes)
    hip6.set_tolerances(1e-10, 1e-2, 1)
    out, dt = {}, {}
    for mode in (0, -1):                                  # eager, then the default (graph: no communicator attached)
        g = ShardGroup(hip6, case6, 2)
        g.set_option("shard_graph", mode)
        g.set_nsteps(20)
        sq, sf = g.alloc(2)
        g.upload(sq, *q)
        res = []
        for rep in range(4):                              # budgets shrink after the first maps: graphs are re-captured
            t0 = time.time(); g.matvec(sf, sq, 0); g.norm(sf); t1 = time.time()
            res.append(np.concatenate([a.ravel() for a in g.download(sf)]))
            g.copy(sq, sf); g.scal(sq, 1.0 / g.norm(sq))
        out[mode], dt[mode] = res, t1 - t0
        g.free([sq, sf]); g.close()
    for a, b in zip(out[0], out[-1]):
        assert np.array_equal(a, b)
    print("sharded step, 2 virtual ranks, 20 steps: eager %.1f ms, graph %.1f ms" % (1e3 * dt[0], 1e3 * dt[-1]))
    hip6.set_nsteps(100)


def test_release_parent_leaves_working_shards(case6, oracle6_nosolve, modes):
    """nsk_shard_release_parent: the shards keep computing (bit-identical map) once the full-mesh context has given its
    element-major device arrays back; the released parent refuses to compute."""
    import torch
    from nekstab_amd.capi import NekStabHip, NskError
    from nekstab_amd.sharded import ShardGroup
    q = _mode(oracle6_nosolve, modes)
    free0 = torch.cuda.mem_get_info()[0]
    full = NekStabHip(case6, case6.meta["vert"], case6.meta["nvert"], tol_helm=1e-10, tol_pres=1e-2, tol_relative=1,
                      max_helm_iter=120, max_pres_iter=48)
    g = ShardGroup(full, case6, 2)
    g.set_nsteps(6)
    sq, sf = g.alloc(2)
    g.upload(sq, *q)
    g.matvec(sf, sq, 0)
    before = np.concatenate([a.ravel() for a in g.download(sf)])
    free1 = torch.cuda.mem_get_info()[0]
    g.release_parent()
    free2 = torch.cuda.mem_get_info()[0]
    g.matvec(sf, sq, 0)
    after = np.concatenate([a.ravel() for a in g.download(sf)])
    assert np.array_equal(before, after)
    with pytest.raises(NskError):
        full.alloc(1)
    print("device memory: parent + 2 shards %.0f MB, after release %.0f MB" % ((free0 - free1) / 2**20, (free0 - free2) / 2**20))
    assert free2 > free1
    g.free([sq, sf]); g.close(); full.close()
