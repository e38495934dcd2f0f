"""a13 and the dense half of a12 against THE REFERENCE ITSELF: core/lapack_wrapper.f compiled unchanged (oracle/build_ref.py ->
oracle/_ref/liblapack_wrapper_ref.so) -- its eig (dgeev + pair assembly + sort, :129-251), schur / select_eigvals (:7-59,
:258-270) and ordschur (:70-122) -- compared with nekstab_amd/krylov.py and host/krylov_host.f90 on Hessenberg matrices of a
config-2 Krylov-Schur run (tests/golden/cfg2_hessenberg.npz, two restarts) and on a synthetic one.  CPU only."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref
    path = build_ref.build()
    if path is None:
        pytest.skip("oracle/_ref/liblapack_wrapper_ref.so is not built and cannot be built here (no flang or no /root/reference)")
    return RefWrapper(C.CDLL(path))


class RefWrapper:
    """ctypes view of the reference's Fortran entry points (everything by reference, column-major, default integer 4 bytes)."""
    def __init__(self, lib):
        self.lib = lib
        lib.select_eigvals_.restype = C.c_int

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def eig(self, A):
        n = A.shape[0]
        Af = np.asfortranarray(A, dtype=np.float64).copy(order="F")
        vecs = np.zeros((n, n), dtype=np.complex128, order="F")
        vals = np.zeros(n, dtype=np.complex128)
        self.lib.eig_(self._p(Af), self._p(vecs), self._p(vals), C.byref(C.c_int(n)))
        return vals, vecs

    def schur(self, A):
        n = A.shape[0]
        T = np.asfortranarray(A, dtype=np.float64).copy(order="F")
        Z = np.zeros((n, n), order="F")
        vals = np.zeros(n, dtype=np.complex128)
        self.lib.schur_(self._p(T), self._p(Z), self._p(vals), C.byref(C.c_int(n)))
        return T, Z, vals

    def ordschur(self, T, Z, sel):
        n = T.shape[0]
        T2, Z2 = T.copy(order="F"), Z.copy(order="F")
        s = np.asarray(sel, dtype=np.int32).copy()
        self.lib.ordschur_(self._p(T2), self._p(Z2), self._p(s), C.byref(C.c_int(n)))
        return T2, Z2

    def select_eigvals(self, wr, wi):
        return bool(self.lib.select_eigvals_(C.byref(C.c_double(wr)), C.byref(C.c_double(wi))))


def _synthetic(k=30, seed=11):
    import scipy.linalg as sla
    rng = np.random.default_rng(seed)
    blocks, i = np.zeros((k, k)), 0
    for z in (1.02 * np.exp(0.7j), 0.95 + 0j, 0.93 * np.exp(0.3j)):
        if z.imag == 0:
            blocks[i, i] = z.real; i += 1
        else:
            blocks[i:i + 2, i:i + 2] = [[z.real, z.imag], [-z.imag, z.real]]; i += 2
    while i + 1 < k:
        z = 0.85 * rng.random() * np.exp(2j * np.pi * rng.random())
        blocks[i:i + 2, i:i + 2] = [[z.real, z.imag], [-z.imag, z.real]]; i += 2
    if i < k:
        blocks[i, i] = 0.4
    X = rng.standard_normal((k, k))
    Hk = sla.hessenberg(X @ blocks @ np.linalg.inv(X))
    H = np.zeros((k + 1, k))
    H[:k] = Hk
    H[k, k - 1] = 0.37
    return H


def _inputs():
    out = [("synthetic k=30", _synthetic(), 3, 0.10)]
    p = os.path.join(GOLDEN, "cfg2_hessenberg.npz")
    if os.path.exists(p):
        z = np.load(p)
        for i, H in enumerate(z["H_restart"]):
            out.append(("config 2, restart %d (k=%d)" % (i + 1, int(z["k_dim"])), H, int(z["schur_tgt"]), float(z["schur_del"])))
        out.append(("config 2, final H", z["H_final"], int(z["schur_tgt"]), float(z["schur_del"])))
    return out


def test_config2_fixture_is_committed():
    assert os.path.exists(os.path.join(GOLDEN, "cfg2_hessenberg.npz")), "tests/golden/cfg2_hessenberg.npz (make_hessenberg_fixture.py) is missing"
    z = np.load(os.path.join(GOLDEN, "cfg2_hessenberg.npz"))
    assert len(z["H_restart"]) >= 1, "the fixture must hold at least one Krylov-Schur restart"


def test_select_eigvals_is_the_0p9_circle(ref):
    """core/lapack_wrapper.f:258-270 against the predicate krylov.schur_condensation hands to dgees."""
    for wr, wi in [(0.9, 0.0), (0.9000000001, 0.0), (0.6363961, 0.6363961), (0.64, 0.64), (-0.95, 0.0), (0.0, 0.0), (0.3, -0.86), (0.0, 0.9), (0.0, -0.91)]:
        assert ref.select_eigvals(wr, wi) == bool(np.hypot(wr, wi) > 0.9), (wr, wi)


@pytest.mark.parametrize("which", range(4))
def test_eig_of_the_host_equals_the_references(ref, which):
    """`eig`: the same Ritz values in the same order (decreasing modulus), the same residual estimates |H(k+1,k) y_k|
    (core/eigensolvers.f:349) and the same eigenvectors up to a unit complex factor."""
    from nekstab_amd import krylov
    ins = _inputs()
    if which >= len(ins):
        pytest.skip("input %d needs tests/golden/cfg2_hessenberg.npz" % which)
    name, H, tgt, delta = ins[which]
    k = H.shape[1]
    vr, Vr = ref.eig(H[:k, :k])
    vp, Vp = krylov.eig_sorted(H[:k, :k])
    assert np.all(np.diff(np.abs(vr)) <= 1e-14), "reference order is decreasing modulus"
    assert np.abs(np.abs(vr) - np.abs(vp)).max() < 1e-12
    assert np.abs(np.sort_complex(vr) - np.sort_complex(vp)).max() < 1e-12
    # pair every reference eigenvalue with the host's (conjugates of equal modulus may be swapped inside their pair)
    rr = np.abs(H[k, k - 1] * Vr[k - 1, :]) / np.linalg.norm(Vr, axis=0)
    rp = np.abs(H[k, k - 1] * Vp[k - 1, :]) / np.linalg.norm(Vp, axis=0)
    worst_v = 0.0
    for i in range(k):
        j = int(np.argmin(np.abs(vp - vr[i])))
        assert abs(vp[j] - vr[i]) < 1e-12
        assert abs(rr[i] - rp[j]) < 1e-10 * max(1.0, abs(H[k, k - 1]))
        gap = np.min(np.abs(np.delete(vp, j) - vp[j]))
        if gap > 1e-3:                                   # eigenvector conditioning ~ 1 / gap
            c = abs(np.vdot(Vr[:, i], Vp[:, j])) / (np.linalg.norm(Vr[:, i]) * np.linalg.norm(Vp[:, j]))
            worst_v = max(worst_v, 1.0 - c)
    print(name, "eig: max |vals diff| %.1e, worst 1 - |cos| of eigenvectors %.1e" % (np.abs(np.sort_complex(vr) - np.sort_complex(vp)).max(), worst_v))
    assert worst_v < 1e-9


@pytest.mark.parametrize("which", range(3))
def test_schur_restart_of_the_host_equals_the_references(ref, which, tmp_path):
    """schur + select + ordschur: the reference's dgees / dtrsen wrappers on the same H against krylov.schur_condensation
    (SciPy's LAPACK) and host/krylov_host.f90 (dense_check): same selection set, same kept Ritz values, same kept invariant
    subspace, same coupling row b^T Z up to the orthogonal freedom of a Schur basis (its norm and the Ritz residuals)."""
    from nekstab_amd import krylov
    from tests.dense_backend import DenseBackend, Vec
    ins = _inputs()
    if which >= len(ins) or (which > 0 and "restart" not in ins[which][0]):
        pytest.skip("input %d needs tests/golden/cfg2_hessenberg.npz" % which)
    name, H, tgt, delta = ins[which]
    k = H.shape[1]
    # ---- reference
    T, Z, vals = ref.schur(H[:k, :k])
    assert np.abs(Z.T @ Z - np.eye(k)).max() < 1e-12 and np.abs(Z @ T @ Z.T - H[:k, :k]).max() < 1e-11 * max(1.0, np.abs(H).max())
    sel_r = krylov.select_eigenvalues(vals, delta, tgt)            # (select_eigenvalues lives in core/eigensolvers.f: not compilable; host rule on the reference's values)
    T2, Z2 = ref.ordschur(T, Z, sel_r)
    ms_r = int(sel_r.sum())
    # ---- Python host (on a dense backend: the device part of the restart is the basis rotation)
    be = DenseBackend(np.eye(k + 1))
    Q = [Vec(k + 1) for _ in range(k + 1)]
    for i, q in enumerate(Q):
        q.a[i] = 1.0
    Hp = H.copy()
    mstart = krylov.schur_condensation(be, Q, Hp, k, 1, delta, tgt)
    ms_p = mstart - 1
    assert ms_p == ms_r, (ms_p, ms_r)
    ev_r = np.sort_complex(np.linalg.eigvals(T2[:ms_r, :ms_r]))
    ev_p = np.sort_complex(np.linalg.eigvals(Hp[:ms_p, :ms_p]))
    assert np.abs(ev_r - ev_p).max() < 1e-12
    # the dgees eigenvalues themselves, as sets
    Tp, Zp, _ = __import__("scipy.linalg", fromlist=["schur"]).schur(H[:k, :k], output="real", sort=lambda re, im: np.hypot(re, im) > 0.9)
    assert np.abs(np.sort_complex(vals) - np.sort_complex(krylov._schur_block_eigs(Tp))).max() < 1e-12
    # kept invariant subspace: the rotated basis vectors are the columns of Z (identity basis)
    Zk_p = np.stack([q.a[:k] for q in Q[:ms_p]], axis=1)
    Pr, Pp = Z2[:, :ms_r] @ Z2[:, :ms_r].T, Zk_p @ Zk_p.T
    b = np.zeros(k); b[k - 1] = H[k, k - 1]
    br, bp = (b @ Z2)[:ms_r], Hp[ms_p, :ms_p]
    print(name, "restart: ms %d, |P_ref - P_host| %.1e, |b^T Z| %.6e / %.6e, elementwise |T2 - Hn| %.1e" %
          (ms_r, np.abs(Pr - Pp).max(), np.linalg.norm(br), np.linalg.norm(bp), np.abs(np.abs(T2[:ms_r, :ms_r]) - np.abs(Hp[:ms_p, :ms_p])).max()))
    assert np.abs(Pr - Pp).max() < 1e-10
    assert abs(np.linalg.norm(br) - np.linalg.norm(bp)) < 1e-12 * max(1.0, abs(H[k, k - 1]))
    # Ritz residuals of the kept block (what the next cycle starts from): |b^T Z y_i| for the eigenvectors of the kept block
    def resid(Tk, brow):
        w, Y = np.linalg.eig(Tk)
        o = np.argsort(-np.abs(w) + 1e-3 * np.sign(w.imag))
        return np.abs(brow @ Y[:, o]) / np.linalg.norm(Y[:, o], axis=0), w[o]
    rr, wr_ = resid(T2[:ms_r, :ms_r], br)
    rp, wp_ = resid(Hp[:ms_p, :ms_p], bp)
    assert np.abs(np.sort(rr) - np.sort(rp)).max() < 1e-10 * max(1.0, abs(H[k, k - 1]))
    # ---- Fortran host (flang), when its dense driver is built or buildable
    exe = os.path.join(ROOT, "host", "dense_check")
    if not os.path.exists(exe) and shutil.which("flang"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "dense_check"], check=True, stdout=subprocess.DEVNULL)
    if os.path.exists(exe):
        fin, fout = str(tmp_path / "H.bin"), str(tmp_path / "out.bin")
        np.asfortranarray(H).T.ravel().tofile(fin)
        out = subprocess.run([exe, fin, fout, str(k), str(tgt), str(delta)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        raw = np.fromfile(fout)
        vals_f = raw[:2 * k:2] + 1j * raw[1:2 * k:2]
        o = 2 * k + 2 * k * k
        ms_f = int(raw[o]); o += 1
        Hn_f = raw[o:o + (k + 1) * k].reshape(k, k + 1).T; o += (k + 1) * k
        Z_f = raw[o:o + k * k].reshape(k, k).T
        vr, _ = ref.eig(H[:k, :k])
        assert np.abs(np.abs(vals_f) - np.abs(vr)).max() < 1e-12 and np.abs(np.sort_complex(vals_f) - np.sort_complex(vr)).max() < 1e-12
        assert ms_f == ms_r
        assert np.abs(np.sort_complex(np.linalg.eigvals(Hn_f[:ms_f, :ms_f])) - ev_r).max() < 1e-12
        Pf = Z_f[:, :ms_f] @ Z_f[:, :ms_f].T
        assert np.abs(Pf - Pr).max() < 1e-10
        print(name, "Fortran host: ms %d, |P_ref - P_fortran| %.1e" % (ms_f, np.abs(Pf - Pr).max()))
