"""BASELINE configs 3, 4 and 5 AT THEIR FULL SIZE on one GPU: a handful of time steps each, checked
through size-independent properties -- every inner solve converges within its budget, the discrete divergence of the
result is small against that of the input (the pressure projection does its work), the energy stays bounded -- and
the time per step is printed.  The eigenproblems of these configurations are multi-GPU runs (bench.py --gpus N)."""
import os
import time

import numpy as np
import pytest

from tests.conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _div_norm_2d(h, u, v):
    return float(np.sqrt(np.sum(h.t_opdiv(u, v) ** 2)))


def test_config3_cylinder_lx1_12_E7984():
    """cylinder, 2 x 2 refined mesh (E = 7984), lx1 = 12, lxd = 18: 1.15 M points per field, 861 steps per matvec."""
    from nekstab_amd import mesh, seed
    from nekstab_amd.settings import production_context
    case = mesh.refine_case_2x2(mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 12))
    assert case.nel == 7984 and case.lx1 == 12
    t0 = time.perf_counter()
    h = production_context(case)
    setup = time.perf_counter() - t0
    assert h.nvel == 7984 * 144 and h.nsteps > 800
    qx, qy = seed.add_noise(case)
    q, f = h.alloc(2)
    h.upload(q, qx, qy, np.zeros(h.npres))
    h.scal(q, 1.0 / h.norm(q))
    nst = 8
    h.set_nsteps(nst)
    h.matvec(f, q, 0)                                     # first map: launch budgets / graphs settle
    t0 = time.perf_counter(); h.matvec(f, q, 0); h.norm(f); dt = time.perf_counter() - t0
    st = h.stats()
    a, b = h.download(q), h.download(f)
    d_in, d_out = _div_norm_2d(h, a[0], a[1]), _div_norm_2d(h, b[0], b[1])
    print("config 3: E %d lx1 12, %d points/field, nsteps/matvec %d, set-up %.0f s, %.2f ms per time step (%.1f Helmholtz + %.1f pressure iterations), "
          "|div| %.2e -> %.2e, |f| %.3f" % (case.nel, h.nvel, 861, setup, 1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst, d_in, d_out, h.norm(f)))
    assert st["unconverged"] == 0 and st["steps"] == nst
    assert d_out < 1e-3 * d_in
    assert 0.0 < h.norm(f) < 1.0                           # noise decays under the linearised operator over 8 steps
    h.close()


def test_config4_backstep_extruded_E50100_adjoint():
    """back_fstep extruded over 30 spanwise layers (E = 50 100 hexahedra, lx1 = 8): 25.65 M points per field, adjoint map
    (all-Dirichlet / periodic velocity: singular pressure operator), Chebyshev coarse solve on 51 k vertices."""
    from nekstab_amd import mesh, mesh3d
    from nekstab_amd.capi import NekStabHip
    c2 = mesh.load_case_npz(os.path.join(GOLDEN, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
    nz = 30
    c3 = mesh3d.extrude_case(c2, nz, 6.0, periodic=True)
    assert c3.nel == 50100
    t0 = time.perf_counter()
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=48)
    setup = time.perf_counter() - t0
    assert h.nvel == 50100 * 512 and h.nstate == 3 * h.nvel + h.npres
    rng = np.random.default_rng(1)
    tg = np.load(os.path.join(GOLDEN, "backstep_tg.npz"))
    u2 = mesh.interp_field_2d(tg["pRe_u"].astype(np.float64), 8) * c2.mask
    w = 1e-2 * np.sin(2 * np.pi * c3.z / 6.0) * c3.mask * np.abs(mesh3d.extrude_field(u2[0], nz))      # three-dimensional, C0, satisfies the BCs
    q, f = h.alloc(2)
    h.upload3(q, mesh3d.extrude_field(u2[0], nz), mesh3d.extrude_field(u2[1], nz), w, np.zeros(h.npres))
    h.scal(q, 1.0 / h.norm(q))
    nst = 3
    h.set_nsteps(nst)
    h.matvec(f, q, 1)
    t0 = time.perf_counter(); h.matvec(f, q, 1); nf = h.norm(f); dt = time.perf_counter() - t0
    st = h.stats()
    a, b = h.download3(q), h.download3(f)
    d_in = float(np.sqrt(np.sum(h.t_op3(1, np.stack(a[:3])) ** 2)))
    d_out = float(np.sqrt(np.sum(h.t_op3(1, np.stack(b[:3])) ** 2)))
    print("config 4: E %d lx1 8, %d points/field (state %.0f MB), set-up %.0f s, %.1f ms per time step (%.1f Helmholtz + %.1f pressure iterations), "
          "|div| %.2e -> %.2e, |f| %.4f" % (c3.nel, h.nvel, 8e-6 * h.nstate, setup, 1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst, d_in, d_out, nf))
    assert st["unconverged"] == 0
    assert d_out < 0.2 * d_in                              # a 1e-2 pressure tolerance per step, input with O(1) divergence in w
    assert 0.5 < nf < 2.0
    h.close()


def test_config4_rank_local_setup_and_shards_at_full_size():
    """Config 4 is an 8-GPU case: at its full size, two ranks set up from THEIR sub-meshes (own elements + two rings:
    sharded.LocalParent), exchange volume / CFL maximum / coarse rows, and run the sharded adjoint step with host-checked
    convergence -- equal to the whole-mesh single-rank map, the block-circulant coarse solve found on the gathered rows."""
    from nekstab_amd import mesh, mesh3d
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardGroup, local_parents, partition_rcb
    c2 = mesh.load_case_npz(os.path.join(GOLDEN, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
    nz = 30
    c3 = mesh3d.extrude_case(c2, nz, 6.0, periodic=True)
    kw = dict(tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=96)
    tg = np.load(os.path.join(GOLDEN, "backstep_tg.npz"))
    u2 = mesh.interp_field_2d(tg["pRe_u"].astype(np.float64), 8) * c2.mask
    w = 1e-2 * np.sin(2 * np.pi * c3.z / 6.0) * c3.mask * np.abs(mesh3d.extrude_field(u2[0], nz))
    q = (mesh3d.extrude_field(u2[0], nz), mesh3d.extrude_field(u2[1], nz), w, np.zeros((c3.nel, 6, 6, 6)))
    R, nst = 2, 3
    part = partition_rcb(c3, R)
    t0 = time.perf_counter()
    P, _ = local_parents(c3, R, part, **kw)
    t_local = time.perf_counter() - t0
    try:
        assert all(p.nel < 0.62 * c3.nel for p in P)
        g = ShardGroup(P, c3, R, part)
        g.release_parent()
        g.set_option("shard_hostcheck", 1)
        g.set_nsteps(nst)
        sq, sf = g.alloc(2)
        g.upload3(sq, *q)
        g.matvec(sf, sq, 1)
        got = g.download3(sf)
        its = g.stats()
        g.close()
    finally:
        for p in P:
            p.close()
    t0 = time.perf_counter()
    h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], **kw)
    t_whole = time.perf_counter() - t0
    try:
        assert P[0].nsteps == h.nsteps and abs(P[0].dt - h.dt) < 1e-15
        h.set_nsteps(nst)
        vq, vf = h.alloc(2)
        h.upload3(vq, *q)
        h.matvec(vf, vq, 1)
        ref = h.download3(vf)
        st = h.stats()
        sc = max(np.abs(ref[k]).max() for k in range(3))
        err = max(np.abs(got[k] - ref[k]).max() for k in range(3)) / sc
        print("config 4 on two rank-local shards: %d + %d elements of %d, set-up of both ranks %.0f s (whole mesh %.0f s), velocity difference %.1e, "
              "pressure iterations %d / %d" % (P[0].nel, P[1].nel, c3.nel, t_local, t_whole, err, its["pres_iters"], st["pres_iters"]))
        assert err < 1e-10
        assert abs(its["pres_iters"] - st["pres_iters"]) <= 2 and its["unconverged"] == 0
    finally:
        h.close()


def test_config5_lid_driven_cube_E99452_lx1_10():
    """lid-driven cube, 46 x 46 x 47 = 99 452 hexahedra with wall clustering as examples/lid_driven/cav.box, lx1 = 10
    (lxd = 15): 99.5 M points per field, 349 M unknowns (2.8 GB per state vector), ~150 GB of device memory on ONE GPU;
    closed domain (singular pressure operator), Chebyshev coarse solve on 103 823 vertices.  Full equations (the Newton-
    Krylov map of config 5) and the linearised map, two time steps each."""
    from nekstab_amd import mesh3d
    from nekstab_amd.capi import NekStabHip
    stretch = lambda xi: 0.5 * (1.0 - np.cos(np.pi * xi))              # Chebyshev-like wall clustering
    lid = lambda x, y, z: np.stack([(1.0 - (2.0 * x - 1.0) ** 2) ** 2 * (1.0 - (2.0 * z - 1.0) ** 2) ** 2 * (y > 1.0 - 1e-12), 0 * x, 0 * x])
    t0 = time.perf_counter()
    c = mesh3d.box_case_3d(46, 46, 47, 10, lengths=(1.0, 1.0, 1.0), re=1000.0, endtime=0.02, stretch=stretch, ub_func=lid)
    # a smooth divergence-free-ish interior base flow (one big vortex in x-y) so that the CFL rule sees a real velocity
    X, Y, Z = c.x, c.y, c.z
    sx, sy, sz = np.sin(np.pi * X), np.sin(np.pi * Y), np.sin(np.pi * Z)
    c.ub[0] = sx ** 2 * np.sin(2 * np.pi * Y) * sz ** 2 * c.mask
    c.ub[1] = -np.sin(2 * np.pi * X) * sy ** 2 * sz ** 2 * c.mask
    del X, Y, Z, sx, sy, sz
    assert c.nel == 99452
    tm = time.perf_counter() - t0
    t0 = time.perf_counter()
    # cav.box's Chebyshev clustering at 46 elements per direction means cell-size ratios of 29: the pressure GMRES needs ~50
    # iterations per step here, i.e. a restart (cycle of 48: k_gmres_restart)
    h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-9, tol_pres=1e-2, tol_relative=1, max_helm_iter=200, max_pres_iter=192)
    setup = time.perf_counter() - t0
    assert h.nvel == 99452 * 1000 and h.nstate == 3 * h.nvel + h.npres
    q, f = h.alloc(2)
    rng = np.random.default_rng(2)
    w = 1e-2 * rng.standard_normal(c.x.shape) * c.mask
    h.upload3(q, c.ub[0] + w, c.ub[1] - w, w, np.zeros(h.npres))
    del w
    n0 = h.norm(q)
    nst = 2                                            # (two time steps per map: a time step takes a second at this size)
    h.set_nsteps(nst)
    out = {}
    for name, run in (("linearised", lambda: h.matvec(f, q, 0)), ("full equations", lambda: h.nonlinear_map(f, q))):
        run()
        t0 = time.perf_counter(); run(); nf = h.norm(f); dt = time.perf_counter() - t0
        st = h.stats()
        out[name] = (1e3 * dt / nst, st["helm_iters"] / nst, st["pres_iters"] / nst, nf)
        assert st["unconverged"] == 0 and np.isfinite(nf) and 0.5 * n0 < nf < 2.0 * n0
    print("config 5: E %d lx1 10, %d points/field, state %.2f GB, mesh %.0f s, set-up %.0f s; " % (c.nel, h.nvel, 8e-9 * h.nstate, tm, setup) +
          "; ".join("%s %.0f ms per time step (%.1f Helmholtz + %.1f pressure iterations)" % (k, v[0], v[1], v[2]) for k, v in out.items()))
    h.close()


def test_bench_case_cfg4_record_shape():
    """`bench.py --case cfg4` (BASELINE configs[3], the bandwidth-sized configuration) on a thin slab of the same mesh: one JSON
    line with the hexahedral byte accounting (Gram-Schmidt bytes from the logged basis indices, coarse-solve bytes from the
    context), the dominant-kernel and end-to-end rooflines and the live per-kernel table."""
    import json, subprocess, sys
    env = dict(os.environ, NSK_BENCH_CFG4_LAYERS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--case", "cfg4", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert "BASELINE configs[3]" in r["config"]["workload"] and "adjoint" in r["config"]["workload"] and r["dtype"] == "f64"
    assert r["value"] > 0 and r["steps"] == 1 and r["n_gpus"] == 1 and r["capped_solves"] == 0      # (a thin slab runs captured graphs with launch budgets: a redone first map is normal there)
    per = r["bytes_per_matvec"]["per_time_step_by_kernel"]
    assert per["K7 gram-schmidt (x n_pres)"] > 0 and per["K6 coarse (x n_pres)"] > 0 and r["coarse_bytes_per_solve"] > 0
    assert r["pres_basis_index_sum_per_step"] > 0
    assert r["bytes_per_matvec"]["time_stepper_shared_arrays_once"] < r["bytes_per_matvec"]["time_stepper"]
    rf = r["roofline"]
    # `frac` = every distinct array of the launch once; the SURVEY's per-component rule (shared arrays counted three times) is kept as `survey_rule`
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1.0 and rf["frac"] < rf["survey_rule"]["frac"] < 1.5 and rf["peak"] == 8000.0
    assert 0 < r["roofline_end_to_end"]["frac_shared_arrays_once"] < r["roofline_end_to_end"]["frac"] < 1.0
    assert set(r["kernels"]) >= {"helm", "divgs", "schwarz", "gs_dots8", "gs_lag8"} and all(v["avg_us"] > 0 for v in r["kernels"].values())
    print({k: (round(v["avg_us"], 1), round(v["frac"], 2)) for k, v in r["kernels"].items()}, r["ms_per_time_step"])
