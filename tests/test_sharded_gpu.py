"""Element sharding with virtual ranks on one GPU: the sharded time-stepper (dssum halo,
Schwarz-overlap halo, all-reduced dot products and coarse restriction) reproduces the single-rank
result, and the host Arnoldi runs unchanged on sharded vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    s.free([a, b]); s.close()
    hip6.free([vq, vf]); hip6.set_nsteps(100)
