"""Krylov vector algebra and the end-to-end Arnoldi on the device, through the C-ABI."""
import os

import numpy as np
import pytest

from nekstab_amd import krylov

pytestmark = pytest.mark.gpu


def _rand_state(case, rng):
    return (rng.standard_normal(case.x.shape), rng.standard_normal(case.x.shape),
            rng.standard_normal((case.nel, case.lx1 - 2, case.lx1 - 2)))


def test_inner_product_and_blas1(hip6, oracle6_nosolve, case6):
    o = oracle6_nosolve
    rng = np.random.default_rng(5)
    a, b = _rand_state(case6, rng), _rand_state(case6, rng)
    va, vb = hip6.alloc(2)
    hip6.upload(va, *a); hip6.upload(vb, *b)
    ref = o.inner(a, b)                       # bm1s-weighted, velocity only (core/krylov_subspace.f:37-38)
    assert abs(hip6.dot(va, vb) - ref) < 1e-12 * abs(ref) + 1e-12
    assert abs(hip6.norm(va) - np.sqrt(o.inner(a, a))) < 1e-12 * np.sqrt(o.inner(a, a))
    hip6.axpy(va, -0.75, vb)                  # krylov_sub2 / add2 with a coefficient, pressure included
    hip6.scal(va, 2.0)
    got = hip6.download(va)
    for g, x, y in zip(got, a, b):
        assert np.abs(g - 2.0 * (x - 0.75 * y)).max() < 1e-13
    hip6.zero(va)
    assert all(np.all(g == 0) for g in hip6.download(va))
    hip6.free([va, vb])


def test_orth_matches_two_pass_gram_schmidt(hip6, oracle6_nosolve, case6):
    o = oracle6_nosolve
    rng = np.random.default_rng(6)
    w = o.bm1s()
    vecs = [_rand_state(case6, rng) for _ in range(5)]
    # orthonormalise the first four on the host (reference algorithm), then compare step 5
    Q = []
    for v in vecs[:4]:
        v = list(v)
        for _ in range(2):
            for q in Q:
                c = o.inner(v, q, w)
                v = [x - c * y for x, y in zip(v, q)]
        n = np.sqrt(o.inner(v, v, w))
        Q.append([x / n for x in v])
    f = list(vecs[4])
    h = np.zeros(4)
    for _ in range(2):                        # update_hessenberg_matrix: two MGS passes
        for i, q in enumerate(Q):
            c = o.inner(f, q, w)
            f = [x - c * y for x, y in zip(f, q)]
            h[i] += c
    beta = np.sqrt(o.inner(f, f, w))
    dv = hip6.alloc(5)
    for d, q in zip(dv[:4], Q):
        hip6.upload(d, *q)
    hip6.upload(dv[4], *vecs[4])
    hg, bg = hip6.orth(dv[4], dv[:4])
    assert np.abs(hg - h).max() < 1e-11 * max(1.0, np.abs(h).max())
    assert abs(bg - beta) < 1e-11 * beta
    got = hip6.download(dv[4])
    for g, x in zip(got, f):
        assert np.abs(g - x / beta).max() < 1e-11
    hip6.free(dv)


def test_basis_gemm_and_gemv(hip6, case6):
    rng = np.random.default_rng(8)
    k = 6
    vs = [_rand_state(case6, rng) for _ in range(k)]
    dv = hip6.alloc(k + 2)
    for d, v in zip(dv, vs):
        hip6.upload(d, *v)
    y = rng.standard_normal(k) + 1j * rng.standard_normal(k)
    hip6.basis_gemv(dv[:k], y, dv[k], dv[k + 1])
    re, im = hip6.download(dv[k]), hip6.download(dv[k + 1])
    for c in range(3):
        M = np.stack([v[c] for v in vs], axis=-1)
        assert np.abs(re[c] - M @ y.real).max() < 1e-12
        assert np.abs(im[c] - M @ y.imag).max() < 1e-12
    Z = rng.standard_normal((k, k))
    hip6.basis_gemm(dv[:k], Z)                # Q <- Q Z  (core/eigensolvers.f:466-474)
    for j in range(k):
        got = hip6.download(dv[j])
        for c in range(3):
            M = np.stack([v[c] for v in vs], axis=-1)
            assert np.abs(got[c] - M @ Z[:, j]).max() < 1e-12
    hip6.free(dv)


@pytest.mark.parametrize("k", [5, 33, 130])
def test_basis_gemm_mfma_sizes(hip6, case6, k):
    """nsk_basis_gemm runs on v_mfma_f64_16x16x4_f64; asymmetric Z, sizes that exercise every
    column-tile template and the k % 4 / k % 16 tails."""
    rng = np.random.default_rng(k)
    vs = [(rng.standard_normal(case6.x.shape), rng.standard_normal(case6.x.shape),
           rng.standard_normal((case6.nel, 4, 4))) for _ in range(k)]
    dv = hip6.alloc(k)
    for d, v in zip(dv, vs):
        hip6.upload(d, *v)
    Z = rng.standard_normal((k, k))
    hip6.basis_gemm(dv, Z)
    for j in (0, 1, k // 2, k - 1):
        got = hip6.download(dv[j])
        for c in range(3):
            ref = sum(vs[q][c] * Z[q, j] for q in range(k))
            assert np.abs(got[c] - ref).max() < 1e-11 * np.sqrt(k)
    hip6.free(dv)


def test_eigen_relation_of_reference_mode(hip6, oracle6_nosolve, modes, spectre):
    """KAT without the oracle in the loop: M(dRe + i dIm) = mu (dRe + i dIm) with mu from the
    reference's Spectre_Hd.dat and the reference's own eigenmode files (fp32)."""
    o = oracle6_nosolve
    J = o.J12
    mu = complex(spectre["Hd"][0, 0], spectre["Hd"][0, 1])
    q = {}
    for k in ("dRe", "dIm"):
        u = modes[k + "_u"].astype(np.float64)
        q[k] = (u[0], u[1], J @ modes[k + "_p"].astype(np.float64) @ J.T)
    hip6.set_tolerances(1e-11, 1e-2, 1)
    hip6.set_nsteps(100)
    vr, vi, fr, fi = hip6.alloc(4)
    hip6.upload(vr, *q["dRe"]); hip6.upload(vi, *q["dIm"])
    assert abs(hip6.dot(vr, vr) + hip6.dot(vi, vi) - 1.0) < 1e-6      # mode normalisation pin
    hip6.matvec(fr, vr, 0); hip6.matvec(fi, vi, 0)
    ray = (hip6.dot(vr, fr) + hip6.dot(vi, fi)) + 1j * (hip6.dot(vr, fi) - hip6.dot(vi, fr))
    assert abs(ray - mu) < 5e-7                                        # 7-digit table
    hip6.free([vr, vi, fr, fi])


def test_arnoldi_leading_pair_matches_reference_table(hip6, case6, spectre):
    """k-step Arnoldi on the device (seed = nekStab add_noise): the leading Ritz pair reproduces
    Spectre_Hd.dat:1-2 (0.7387113 +- 0.6972442 i) within the stated 5e-6."""
    from nekstab_amd import seed
    hip6.set_tolerances(1e-11, 1e-2, 1)
    hip6.set_nsteps(100)
    qx, qy = seed.add_noise(case6)
    v0, v1 = hip6.alloc(2)
    hip6.upload(v0, qx, qy, np.zeros(hip6.npres))
    hip6.scal(v0, 1.0 / hip6.norm(v0))
    hip6.matvec(v1, v0, 0)                    # the reference seeds with M * noise (core/eigensolvers.f:234)
    res = krylov.krylov_schur(hip6, v1, 200, schur_tgt=0)      # k_dim = 200 as in 1cyl.par:8
    mu = complex(spectre["Hd"][0, 0], spectre["Hd"][0, 1])
    lead = res.vals[np.argmin(np.abs(res.vals - mu))]
    print("leading Ritz value", lead, "residual", res.residual[np.argmin(np.abs(res.vals - mu))], "wall", res.wall)
    assert abs(lead - mu) < 5e-6
    lam = krylov.log_transform(np.array([lead]), 1.0)[0]
    assert abs(lam - complex(*spectre["NSd_conv"][0])) < 1e-5
    # Every converged row of the reference table is reproduced.  The leading pair and the real mode
    # 0.9166919 agree to the table's 7 digits; the sub-dominant wake branch is far more sensitive to the
    # inner-solver tolerances (it moves by 1e-3 when ours are loosened to 1e-8/0.3, by 1e-6 when tightened,
    # scripts/compare_spectrum.py) and the reference ran with loose absolute tolerances: <= 1e-4 there.
    ref = spectre["Hd"]
    ok = 0
    for n, r in enumerate(ref[ref[:, 2] < 1e-8]):
        z = complex(r[0], r[1])
        j = np.argmin(np.abs(res.vals - z))
        if res.residual[j] < 1e-7:
            assert abs(res.vals[j] - z) < (5e-6 if n < 3 else 1e-4), (z, res.vals[j])
            ok += 1
    print("matched converged reference rows:", ok)
    assert ok >= 9
    # outpost_ks: files in the reference's formats; eigenmode 1 equals the reference's dRe/dIm up to phase
    import tempfile
    from nekstab_amd import nekio, outpost
    from tests.conftest import make_oracle
    with tempfile.TemporaryDirectory() as td:
        files = outpost.outpost_ks(hip6, res, case6, td, evop="d", eigen_tol=1e-6, maxmodes=4)
        assert os.path.basename(files[0]) == "dRe1cyl0.f00001" and os.path.basename(files[1]) == "dIm1cyl0.f00001"
        tab = nekio.read_spectre(os.path.join(td, "Spectre_Hd.dat"))
        assert tab.shape == (200, 3) and abs(complex(tab[0, 0], tab[0, 1]) - mu) < 5e-6 or abs(complex(tab[0, 0], -tab[0, 1]) - mu) < 5e-6
        fr, fi = nekio.read_fld(files[0]), nekio.read_fld(files[1])
        assert fr.istep == 101 and fr.rdcode == "XUP" and fr.wdsize == 4
    o = make_oracle(case6, build_solvers=False)
    w = o.bm1s()
    ours = (fr.u[0, :, 0] + 1j * fi.u[0, :, 0], fr.u[1, :, 0] + 1j * fi.u[1, :, 0])
    mods = np.load(os.path.join(os.path.dirname(__file__), "golden", "cylinder_modes.npz"))
    refm = (mods["dRe_u"][0] + 1j * mods["dIm_u"][0], mods["dRe_u"][1] + 1j * mods["dIm_u"][1])
    if lead.imag < 0:
        ours = tuple(np.conj(a) for a in ours)
    ov = sum(np.sum(np.conj(a) * w * b) for a, b in zip(refm, ours))
    print("mode overlap |<ref,ours>| =", abs(ov))
    assert abs(abs(ov) - 1.0) < 1e-4
    diff = sum(np.sum(w * np.abs(b * np.conj(ov) / abs(ov) - a) ** 2) for a, b in zip(refm, ours))
    assert np.sqrt(diff) < 2e-3                                        # fp32 files, eigen_tol 1e-6 in the reference
    hip6.free(res.Q + [v0, v1])


def test_krylov_schur_restarts_on_device(hip6, case6, spectre):
    """Small Krylov space + Schur condensation (core/eigensolvers.f:395-499) with the basis rotation
    on the matrix cores: converges to the same leading pair as the reference's 200-vector Arnoldi."""
    from nekstab_amd import seed
    hip6.set_tolerances(1e-11, 1e-2, 1)
    hip6.set_nsteps(100)
    qx, qy = seed.add_noise(case6)
    v0, v1 = hip6.alloc(2)
    hip6.upload(v0, qx, qy, np.zeros(hip6.npres))
    hip6.scal(v0, 1.0 / hip6.norm(v0))
    hip6.matvec(v1, v0, 0)
    res = krylov.krylov_schur(hip6, v1, 48, schur_tgt=2, eigen_tol=1e-6, schur_del=0.10, max_restarts=12)
    mu = complex(spectre["Hd"][0, 0], spectre["Hd"][0, 1])
    lead = res.vals[np.argmin(np.abs(res.vals - mu))]
    print("restarts", res.schur_cnt, "matvecs", res.matvecs, "lead", lead, "wall", res.wall)
    assert res.schur_cnt >= 1
    assert abs(lead - mu) < 5e-6
    hip6.free(res.Q + [v0, v1])


def test_checkpoint_restart(hip6, case6, tmp_path):
    """arnoldi_checkpoint + restart (core/eigensolvers.f:802-905, :284-325, core/IO.f:15-60): a
    factorisation resumed from the KRY/HES files of step 5 reproduces the uninterrupted one."""
    from nekstab_amd import checkpoint, seed
    hip6.set_tolerances(1e-11, 1e-2, 1)
    hip6.set_nsteps(10)
    qx, qy = seed.add_noise(case6)
    k = 8
    Q = hip6.alloc(k + 1)
    hip6.upload(Q[0], qx, qy, np.zeros(hip6.npres))
    hip6.scal(Q[0], 1.0 / hip6.norm(Q[0]))
    H = np.zeros((k + 1, k))
    d = str(tmp_path)
    krylov.arnoldi_factorization(hip6, Q, H, 1, k, 0,
                                 log=lambda m, Hm, dt: checkpoint.arnoldi_checkpoint(hip6, case6, Q, Hm, m, d))
    assert os.path.exists(os.path.join(d, "KRY1cyl0.f00009")) and os.path.exists(os.path.join(d, "HES1cyl0008"))
    assert os.path.exists(os.path.join(d, "Spectre_Hd0005.dat"))
    Q2, H2, ms = checkpoint.load_checkpoint(hip6, case6, d, k, 5)
    assert ms == 6 and np.array_equal(H2[:6, :5], H[:6, :5])
    krylov.arnoldi_factorization(hip6, Q2, H2, ms, k, 0)
    assert np.abs(H2 - H).max() < 1e-8 * np.abs(H).max()
    hip6.free(Q + Q2)
    hip6.set_nsteps(100)
