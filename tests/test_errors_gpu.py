"""Error behaviour of the C-ABI: the product path fails loudly, with a status code and a message, where the reference
aborts (core/krylov_subspace.f:53: NaN in an inner product) or where an argument cannot be served -- it never computes on.
Codes: include/nekstab_hip.h (NSK_EINVAL -1, NSK_EHIP -2, NSK_ENAN -3, NSK_ENOCONV -4)."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(tol_helm=1e-10, tol_pres=1e-6, tol_relative=1, max_helm_iter=100, max_pres_iter=48)


def _small_box(lx1=6, **kw):
    from nekstab_amd import mesh3d
    ubf = lambda x, y, z: np.stack([1.0 - 0.3 * y * y, 0.2 * np.cos(x) * y, 0.1 * np.sin(y + z)])
    return mesh3d.box_case_3d(3, 2, 2, lx1, lengths=(2.0, 1.0, 0.8), outflow_xmax=True, re=40.0, endtime=0.05, ub_func=ubf, **kw)


def test_nan_in_an_inner_product_is_an_error(hip6, case6):
    """core/krylov_subspace.f:53: `if (isnan(alpha)) call exitt` -> NSK_ENAN from norm, dot and the Hessenberg update."""
    from nekstab_amd.capi import NskError
    a, b = hip6.alloc(2)
    bad = np.zeros(case6.x.shape)
    bad[3, 2, 2] = np.nan
    zero = np.zeros(case6.x.shape)
    hip6.upload(a, bad, zero, np.zeros(hip6.npres))
    hip6.upload(b, case6.mask * 1.0, zero, np.zeros(hip6.npres))
    for call in (lambda: hip6.norm(a), lambda: hip6.dot(a, b), lambda: hip6.orth(a, [b])):
        with pytest.raises(NskError) as e:
            call()
        assert e.value.code == -3 and "NaN" in str(e.value)
    hip6.free([a, b])


def test_unsupported_order_and_bad_mesh_are_refused(case6):
    from nekstab_amd.capi import NekStabHip, NskError
    # lx1 outside the built kernel sets
    bad = dataclasses.replace(case6, lx1=7, lxd=10, x=np.zeros((case6.nel, 7, 7)), y=np.zeros((case6.nel, 7, 7)), gid=np.zeros((case6.nel, 7, 7), dtype=np.int64),
                              mask=np.ones((case6.nel, 7, 7)), ub=np.zeros((2, case6.nel, 7, 7)), spng=np.zeros((case6.nel, 7, 7)))
    with pytest.raises(NskError) as e:
        NekStabHip(bad, bad.meta["vert"], bad.meta["nvert"], **KW)
    assert e.value.code == -1
    # hexahedra: an element turned inside out (non-positive Jacobian) and a vertex id outside 0..nvert-1
    c = _small_box()
    x = c.x.copy(); x[1] = x[1][:, :, ::-1]                 # r-direction of one element reversed
    with pytest.raises(NskError) as e:
        NekStabHip(dataclasses.replace(c, x=x), c.meta["vert"], c.meta["nvert"], **KW)
    assert e.value.code == -1 and "Jacobian" in str(e.value)
    vert = np.asarray(c.meta["vert"]).copy(); vert[0, 0] = c.meta["nvert"] + 5
    with pytest.raises(NskError) as e:
        NekStabHip(c, vert, c.meta["nvert"], **KW)
    assert e.value.code == -1 and "vertex" in str(e.value)
    # a zero base flow has no CFL time step (core/matvec.f:26-46 divides by it)
    with pytest.raises(NskError) as e:
        NekStabHip(dataclasses.replace(c, ub=np.zeros_like(c.ub)), c.meta["vert"], c.meta["nvert"], **KW)
    assert e.value.code == -1 and "base flow" in str(e.value)


def test_iteration_cap_is_reported_not_passed_over(case6, modes):
    """A solve that cannot reach its tolerance inside the iteration cap ends the map with NSK_ENOCONV (after the launch budgets
    have been grown to the cap) -- never with a silently unconverged field; "pres_cap" is the one sanctioned exception and is
    counted in the statistics."""
    from nekstab_amd.capi import NekStabHip, NskError
    u = modes["dRe_u"].astype(np.float64)
    h = NekStabHip(case6, case6.meta["vert"], case6.meta["nvert"], tol_helm=1e-12, tol_pres=1e-12, tol_relative=1, max_helm_iter=100, max_pres_iter=3)
    try:
        a, b = h.alloc(2)
        h.upload(a, u[0], u[1], np.zeros(h.npres))
        h.set_nsteps(3)
        with pytest.raises(NskError) as e:
            h.matvec(b, a, 0)
        assert e.value.code == -4 and "iteration cap" in str(e.value)
        assert h.stats()["unconverged"] > 0
    finally:
        h.close()


def test_shard_arguments_are_checked(hip6, case6):
    from nekstab_amd.capi import NskError
    from nekstab_amd.sharded import LocalParent, ShardGroup, partition_rcb
    import ctypes as C
    lib = hip6.lib
    part = partition_rcb(case6, 2)
    out = C.c_void_p()
    ip = C.POINTER(C.c_int)
    bad = part.copy(); bad[0] = 7
    assert lib.nsk_shard_create(hip6.ctx, bad.ctypes.data_as(ip), 0, 2, C.byref(out)) == -1 and b"part[0]" in lib.nsk_last_error()
    assert lib.nsk_shard_create(hip6.ctx, part.ctypes.data_as(ip), 2, 2, C.byref(out)) == -1            # rank outside 0..nranks-1
    assert lib.nsk_shard_create(hip6.ctx, np.zeros_like(part).ctypes.data_as(ip), 1, 2, C.byref(out)) == -1 and b"empty shard" in lib.nsk_last_error()
    # composed maps on shards need f != q; shards of a context refuse the full-mesh entry points
    g = ShardGroup(hip6, case6, 2, part)
    a = g.alloc(1)[0]
    with pytest.raises(NskError) as e:
        g.matvec(a, a, 2)
    assert e.value.code == -1
    assert lib.nsk_set_baseflow(g.ctx[0], a.parts[0]) == -1 and b"nsk_group_set_baseflow" in lib.nsk_last_error()
    g.free([a]); g.close()
    # rank-local set-up: a shard cannot be cut before the ranks' rows have been exchanged, and every vertex needs its row
    lp = LocalParent(case6, part, 0, **KW)
    try:
        assert lib.nsk_shard_create_local(lp.ctx, lp.part_sub.ctypes.data_as(ip), lp.sub.ctypes.data_as(C.POINTER(C.c_longlong)), 0, 2, C.byref(out)) == -1
        assert b"nsk_local_finish" in lib.nsk_last_error()
        with pytest.raises(NskError) as e:
            lp.finish(1.0, lp.ctarg, 0.0, lp.npr_own, lp.rows_u, lp.rows_v, lp.rows_a)          # the other rank's rows are missing
        assert e.value.code == -1 and "no row for vertex" in str(e.value)
    finally:
        lp.close()
