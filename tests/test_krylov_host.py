"""Host Arnoldi / Krylov-Schur logic (nekstab_amd/krylov.py) on a dense numpy backend."""
import numpy as np

from nekstab_amd import krylov
from tests.dense_backend import DenseBackend


def _matrix(n=120, seed=3):
    rng = np.random.default_rng(seed)
    # spectrum inside the unit disc with a leading complex pair, like a time-stepper propagator
    d = np.zeros((n, n))
    lead = 0.95 * np.exp(1j * 0.7)
    d[0, 0], d[0, 1], d[1, 0], d[1, 1] = lead.real, lead.imag, -lead.imag, lead.real
    d[2, 2] = 0.9
    for i in range(3, n):
        d[i, i] = 0.8 * rng.random()
    X = rng.standard_normal((n, n))
    return X @ d @ np.linalg.inv(X), lead


def test_arnoldi_relation():
    A, _ = _matrix()
    be = DenseBackend(A, w=np.linspace(0.5, 1.5, A.shape[0]))
    q0 = be.alloc(1)[0]
    q0.a = np.ones(A.shape[0])
    r = krylov.krylov_schur(be, q0, 30, schur_tgt=0)
    Q = np.stack([q.a for q in r.Q], axis=1)
    # M Q_k = Q_{k+1} H and weighted orthonormality
    assert np.abs(A @ Q[:, :30] - Q @ r.H).max() < 1e-10
    G = Q.T @ (be.w[:, None] * Q)
    assert np.abs(G - np.eye(31)).max() < 1e-12


def test_plain_arnoldi_leading_pair():
    A, lead = _matrix()
    be = DenseBackend(A)
    q0 = be.alloc(1)[0]
    q0.a = np.ones(A.shape[0])
    r = krylov.krylov_schur(be, q0, 60, schur_tgt=0)
    assert abs(abs(r.vals[0]) - abs(lead)) < 1e-8
    assert min(abs(r.vals[0] - lead), abs(r.vals[0] - np.conj(lead))) < 1e-8
    assert r.residual[0] < 1e-6
    assert np.all(np.diff(np.abs(r.vals)) <= 1e-12)           # sorted by decreasing modulus


def test_krylov_schur_restart_converges():
    A, lead = _matrix()
    be = DenseBackend(A)
    q0 = be.alloc(1)[0]
    q0.a = np.ones(A.shape[0])
    r = krylov.krylov_schur(be, q0, 16, schur_tgt=2, eigen_tol=1e-9, schur_del=0.10)
    assert r.schur_cnt >= 1                                    # needed restarts with such a small basis
    assert min(abs(r.vals[0] - lead), abs(r.vals[0] - np.conj(lead))) < 1e-8
    # Krylov-Schur relation still holds after condensation: A Q_k = Q_{k+1} H
    Q = np.stack([q.a for q in r.Q], axis=1)
    assert np.abs(A @ Q[:, :16] - Q @ r.H).max() < 1e-8


def test_select_eigenvalues_matches_reference_rule():
    vals = np.array([0.99, 0.5 + 0.5j, 0.5 - 0.5j, 0.3, 0.2, 0.1, 0.05, 0.95j, -0.95j])
    sel = krylov.select_eigenvalues(vals, 0.10, 0)
    assert sel[0] and sel[7] and sel[8]                         # outside the 0.9 circle
    assert sel.sum() >= 4                                       # at least nev+4
    assert sel[1] == sel[2]                                     # pairs are kept together


def test_log_transform():
    mu = np.array([0.7387113 + 0.6972442j])
    lam = krylov.log_transform(mu, 1.0)
    assert abs(lam[0] - (0.01567373 + 0.7565285j)) < 2e-7      # Spectre_NSd_conv.dat:1


def test_band_arnoldi_relation_and_ritz_values():
    """Band (block) Arnoldi with two seeds (nekstab_amd/krylov.py: band_arnoldi) on a dense numpy backend: the band relation
    M Q_k = Q_{k+2} H holds to rounding, the basis is orthonormal in the weighted inner product, and the leading Ritz values
    converge to the eigenvalues of M with residuals that tell the truth."""
    from nekstab_amd import krylov
    from tests.dense_backend import DenseBackend
    rng = np.random.default_rng(5)
    n, k, b = 120, 40, 2
    lam = np.concatenate([[1.05 * np.exp(0.6j), 1.05 * np.exp(-0.6j), 0.97, 0.9 * np.exp(1.1j), 0.9 * np.exp(-1.1j)], 0.6 * rng.random(n - 5)])
    blocks, i = np.zeros((n, n)), 0
    for z in (lam[0], lam[2], lam[3]):
        if z.imag == 0:
            blocks[i, i] = z.real; i += 1
        else:
            blocks[i:i + 2, i:i + 2] = [[z.real, z.imag], [-z.imag, z.real]]; i += 2
    for z in lam[5:]:
        blocks[i, i] = z.real; i += 1
    X = np.eye(n) + 0.3 * rng.standard_normal((n, n)) / np.sqrt(n)
    A = X @ blocks @ np.linalg.inv(X)
    w = 0.5 + rng.random(n)
    be = DenseBackend(A, w)
    seeds = be.alloc(b)
    for s_ in seeds:
        s_.a = rng.standard_normal(n)
    res = krylov.band_arnoldi(be, seeds, k)
    Qm = np.stack([q.a for q in res.Q], axis=1)
    assert np.abs(Qm.T @ (w[:, None] * Qm) - np.eye(k + b)).max() < 1e-10
    assert np.abs(A @ Qm[:, :k] - Qm @ res.H).max() < 1e-10
    assert np.abs(np.tril(res.H, -b - 1)).max() == 0.0            # band Hessenberg
    for z in (lam[0], lam[2], lam[3]):
        j = int(np.argmin(np.abs(res.vals - z)))
        assert abs(res.vals[j] - z) < 1e-6 and res.residual[j] < 1e-5, (z, res.vals[j], res.residual[j])
    # the residual formula is the true residual of the Ritz pair
    y = res.vecs[:, 0]
    x = Qm[:, :k] @ y
    true = A @ x - res.vals[0] * x
    assert abs(np.sqrt(np.sum(w * np.abs(true) ** 2)) - res.residual[0]) < 1e-10
