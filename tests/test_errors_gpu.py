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


def test_non_finite_input_is_reported_by_the_map_and_does_not_poison_the_context(case6):
    """A NaN in the input vector: the MAP returns NSK_ENAN (BENCH_r04: it surfaced one call later, in nsk_orth), on one lane and
    on the lane of a batch that got it; the context resets its solver state, so the next map of a clean vector equals the map
    of a fresh context bit for bit (lag arrays and the projection space carry nothing over)."""
    from nekstab_amd import seed
    from nekstab_amd.capi import NekStabHip, NskError
    kw = dict(tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, max_helm_iter=120, max_pres_iter=192, nproj=8)
    qx, qy = seed.add_noise(case6)
    zp = np.zeros((case6.nel, 4, 4))

    def fresh():
        h = NekStabHip(case6, case6.meta["vert"], case6.meta["nvert"], **kw)
        h.set_nsteps(8)
        return h
    h0 = fresh()
    a, f = h0.alloc(2)
    h0.upload(a, qx, qy, zp)
    h0.matvec(f, a, 0)
    ref = h0.download(f)
    h0.close()
    h = fresh()
    a, b, f, g = h.alloc(4)
    h.upload(a, qx, qy, zp)
    bad = qx.copy(); bad[5, 2, 3] = np.nan
    h.upload(b, bad, qy, zp)
    with pytest.raises(NskError) as e:
        h.matvec(f, b, 0)
    assert e.value.code == -3 and "non-finite" in str(e.value)
    h.matvec(f, a, 0)
    got = h.download(f)
    assert all(np.array_equal(x, y) for x, y in zip(got[:2], ref[:2]))
    with pytest.raises(NskError) as e:                      # lane 1 gets the bad vector
        h.matvec_batch([f, g], [a, b], 0)
    assert e.value.code == -3 and "lane 1" in str(e.value)
    h.matvec_batch([f, g], [a, a], 0)                       # both lanes clean again
    for v in (f, g):
        got = h.download(v)
        err = max(np.abs(x - y).max() / np.abs(y).max() for x, y in zip(got[:2], ref[:2]))
        print("batch after the failed one: lane result vs fresh context, max relative difference %.2e" % err)
        assert err < 1e-4                                   # (lane 0's projection space is warm, lane 1's cold, the fresh context's empty: the pressure tolerance 1e-6 shows at this level)
    h.close()


def test_kernel_timing_hook_leaves_no_trace_in_the_next_map():
    """nsk_bench_kernel runs the step's kernels hundreds of times on the live solver state (bench.py's roofline and --extras);
    the sequence of BENCH_r04 -- every timing back to back, then maps on lane 0 and on a new lane -- produced NaN.  Now the hook
    marks the state dirty and the next map starts from a reset state: finite, equal to a fresh context's map, band Arnoldi
    (nsk_matvec_batch) with bench.py's old second seed included."""
    import os
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.settings import production_context
    from tests.conftest import GOLDEN
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8)
    qx, qy = seed.add_noise(case)
    zp = np.zeros((case.nel, 6, 6))
    h0 = production_context(case)
    a, f = h0.alloc(2)
    h0.upload(a, qx, qy, zp)
    h0.scal(a, 1.0 / h0.norm(a))
    h0.matvec(f, a, 0)
    ref = h0.download(f)
    h0.close()
    h = production_context(case)
    Q = h.alloc(31)
    h.upload(Q[0], qx, qy, zp)
    h.scal(Q[0], 1.0 / h.norm(Q[0]))
    H = np.zeros((31, 30))
    krylov.arnoldi_factorization(h, Q, H, 1, 30, 0, stats={})
    for kn in ("helm", "convect", "rhs", "pres_rhs", "proj_apply", "gmres_update", "schwarz", "divgs2", "pres_update", "vel_update_proj", "proj_update", "update_coarse3"):
        assert h.bench_kernel(kn, 100)["avg_us"] > 0
    f = h.alloc(1)[0]
    h.matvec(f, Q[0], 0)
    got = h.download(f)
    assert all(np.array_equal(x, y) for x, y in zip(got[:2], ref[:2]))          # reset state = fresh context
    for kn in ("helm", "rhs", "pres_update", "vel_update_proj", "proj_update", "update_coarse3"):
        h.bench_kernel(kn, 100)
    for bw in (2, 3):
        sd = h.alloc(bw)
        h.copy(sd[0], Q[0])
        for j in range(1, bw):
            h.upload(sd[j], qy * np.cos(0.2 * j * case.x), qx * np.cos(0.3 * j * case.y), zp)
        rb = krylov.band_arnoldi(h, sd, 24)
        assert np.isfinite(rb.H).all() and np.isfinite(rb.vals).all()
        h.free(rb.Q); h.free(sd)
        for kn in ("pres_update", "proj_update"):
            h.bench_kernel(kn, 100)
    h.close()
