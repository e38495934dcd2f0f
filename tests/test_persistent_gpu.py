"""The persistent velocity solve (nsk_persist.hpp: right-hand side, every CG iteration and the pressure right-hand side in
ONE launch, device-side grid barriers) against the launch-per-iteration form of the same algorithm."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(case, **kw):
    from nekstab_amd.capi import NekStabHip
    return NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-2, tol_relative=1,
                      schwarz_layers=2, max_helm_iter=100, max_pres_iter=48, **kw)


@pytest.mark.parametrize("lx1,mode", [(6, 0), (8, 0), (8, 1)])
def test_fused_equals_launch_per_iteration(lx1, mode):
    """Same arithmetic, same summation orders, same convergence rule => the same fields to rounding (bit-identical when
    the compiler contracts the same multiply-adds), the same iteration counts; no barrier time-outs; with graphs and
    without."""
    from nekstab_amd import mesh, seed
    from tests.conftest import GOLDEN
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), lx1, adjoint=bool(mode))
    h = _ctx(case, nproj=8)
    h.set_option("proj_reset", 1)                # every map starts with an empty pressure projection space: maps are repeatable
    qx, qy = seed.add_noise(case)
    q, f0, f1, f2 = h.alloc(4)
    h.upload(q, qx, qy, np.zeros(h.npres))
    h.scal(q, 1.0 / h.norm(q))
    h.set_nsteps(12)
    h.set_option("fused", 0)
    h.matvec(f0, q, mode)
    s0 = h.stats()
    h.set_option("fused", 1)                     # raises if the grid cannot be resident
    h.matvec(f1, q, mode)
    s1 = h.stats()
    h.set_option("use_graph", 0)
    h.matvec(f2, q, mode)
    a, b, c2 = h.download(f0), h.download(f1), h.download(f2)
    scale = max(np.abs(a[0]).max(), np.abs(a[1]).max())
    err = max(np.abs(x - y).max() for x, y in zip(a[:2], b[:2])) / scale
    errg = max(np.abs(x - y).max() for x, y in zip(b[:2], c2[:2])) / scale
    print("lx1", lx1, "mode", mode, "fused vs launches: max rel diff %.2e; graph vs eager %.2e; helm iters %d / %d, pres iters %d / %d"
          % (err, errg, s0["helm_iters"], s1["helm_iters"], s0["pres_iters"], s1["pres_iters"]))
    assert err < 1e-12 and errg < 1e-12
    assert s1["pres_iters"] == s0["pres_iters"] and s1["unconverged"] == 0
    h.close()


def test_fused_full_map_vs_oracle_lx1_8():
    """Five direct steps at lx1 = 8 (config 2's order) through the persistent kernel against the oracle: the direct-mode
    full-step comparison at this order."""
    from tests.conftest import oracle8_direct, oracle8_five_steps
    case, o = oracle8_direct()                 # (shared with the other test that steps the lx1 = 8 oracle)
    from nekstab_amd.capi import NekStabHip
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-9, tol_relative=1,
                   schwarz_layers=2, max_helm_iter=150, max_pres_iter=160)       # pressure to 1e-9: restarted GMRES cycles
    rng = np.random.default_rng(3)
    # a smooth C0 field: the interpolated base flow plus a masked perturbation of it
    u = case.ub[0] * case.mask * (1.0 + 0.1 * np.sin(case.x)), case.ub[1] * case.mask + 0.05 * case.mask * np.cos(case.y)
    q = (u[0], u[1], rng.standard_normal((case.nel, 6, 6)) * 1e-3)
    h.set_nsteps(5)
    vq, vf = h.alloc(2)
    h.upload(vq, *q)
    for fused in (1, 0):
        h.set_option("fused", fused)
        h.matvec(vf, vq, 0)
        f = h.download(vf)
        ref = oracle8_five_steps(q)
        num = sum(np.sum(o.bm1 * (x - y) ** 2) for x, y in zip(f[:2], ref[:2]))
        den = sum(np.sum(o.bm1 * y ** 2) for y in ref[:2])
        print("fused", fused, "rel L2 vs oracle %.2e" % np.sqrt(num / den))
        assert np.sqrt(num / den) < 1e-9
    h.close()


@pytest.mark.parametrize("fuse2", [1, 0])
def test_persistent_tails_are_bit_identical_to_launch_budgets(fuse2):
    """(fuse2 = 1: the two-launch GMRES iteration of round 6 and its tail k_pres_tail2; 0: the three launches and k_pres_tail.)
    Round 5: persistent tails (k_helm_tail, k_pres_tail: the same kernel bodies in a loop with grid barriers) against the
    launch-budget form on config 2's mesh: identical Hessenberg matrix and vectors, bit for bit -- heads = median counts, heads
    pushed far below the iteration counts so that the tails do most of the iterations, and the default (budgets with the tail as a
    safety net); no barrier time-out, no redone map."""
    import os
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.settings import production_context
    from tests.conftest import GOLDEN
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8)
    qx, qy = seed.add_noise(case)
    zp = np.zeros((case.nel, 6, 6))
    K = 14

    def run(tail, off_h=0, off_p=0):
        h = production_context(case)
        h.set_option("fuse2", fuse2)
        h.set_option("tail", tail)
        h.set_option("tail_off_h", off_h)
        h.set_option("tail_off_p", off_p)
        Q = h.alloc(K + 1)
        h.upload(Q[0], qx, qy, zp)
        h.scal(Q[0], 1.0 / h.norm(Q[0]))
        H = np.zeros((K + 1, K))
        krylov.arnoldi_factorization(h, Q, H, 1, K, 0, stats={})
        last = h.download(Q[K])
        st = h.stats()
        hh, pp = h.step_iters()
        h.close()
        return H, last, st, hh.copy(), pp.copy()
    H0, v0, s0, h0, p0 = run(0, 8, 8)   # (head-room + 8 / + 8: the reference run must not overflow a budget -- a redone map starts from the
    #                                      projection space its failed attempt left: different bits by design)
    print("reference run (launch budgets, no tails): retries", s0["retries"], "budgets per step %.2f / %.2f" % (s0["step_budget_helm_mean"], s0["step_budget_pres_mean"]))
    assert s0["tail_maps"] == 0 and s0["retries"] == 0
    for mode, off in (((1, (0, 0)), (1, (-6, -3)), (2, (0, 0)), (-1, (0, 0))) if fuse2 else ((1, (-6, -3)), (-1, (0, 0)))):       # median heads; heads far below the counts; safety net; the default
        H1, v1, s1, h1, p1 = run(mode, *off)
        print("tail maps", s1["tail_maps"], "heads per step %.2f / %.2f" % (s1["step_budget_helm_mean"], s1["step_budget_pres_mean"]), "retries", s1["retries"],
              "iterations per step %.3f / %.3f" % (s1["total_helm_iters"] / s1["total_steps"], s1["total_pres_iters"] / s1["total_steps"]))
        assert s1["tail_maps"] >= K - 2 and s1["retries"] == 0
        assert np.array_equal(h0, h1) and np.array_equal(p0, p1)           # the same iteration counts, step by step
        assert np.array_equal(H0, H1)
        assert all(np.array_equal(a, b) for a, b in zip(v0, v1))
