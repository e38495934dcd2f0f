"""The Fortran host (flang, iso_c_binding) over the C-ABI: host/arnoldi_host runs a k-step Arnoldi
with nsk_matvec / nsk_orth on the device and LAPACK dgeev on the host; its Hessenberg matrix and
Ritz values must coincide with the Python host driving the same library.
(Runs first among the GPU modules: the child process is started before this process touches the GPU.)"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_fortran_arnoldi_matches_python_host(tmp_path):
    # (`spectre` fixture not needed here)
    exe = os.path.join(ROOT, "host", "arnoldi_host")
    if not os.path.exists(exe):
        if shutil.which("flang") is None:
            pytest.skip("flang not available and host/arnoldi_host not prebuilt")
        subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True)
    from nekstab_amd import casefile, krylov, mesh, seed
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    qx, qy = seed.add_noise(case)
    pr = np.zeros((case.nel, 4, 4))
    cb = str(tmp_path / "case.bin")
    casefile.write_case_bin(cb, case, (qx, qy, pr))
    k = 12
    out = subprocess.run([exe, cb, str(k), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "nsteps = 100" in out.stdout
    Hf = np.array(open(str(tmp_path / "HES.txt")).read().split(), dtype=float)[: (k + 1) * k].reshape(k + 1, k)
    ritz = np.loadtxt(str(tmp_path / "ritz_full.txt"))
    spec = np.loadtxt(str(tmp_path / "Spectre_Hd.dat"))
    assert spec.shape == (k, 3)
    # same factorisation from the Python host, at the SAME settings: the production ones, which the case file carries
    from nekstab_amd.settings import production_context
    h = production_context(case)
    v0 = h.alloc(1)[0]
    h.upload(v0, qx, qy, pr)
    res = krylov.krylov_schur(h, v0, k, schur_tgt=0)
    # iterative inner solves => agreement to the solver tolerance, not bitwise
    assert np.abs(Hf - res.H).max() < 1e-7 * np.abs(res.H).max()
    fv = ritz[:, 0] + 1j * ritz[:, 1]
    assert np.abs(np.sort(np.abs(fv)) - np.sort(np.abs(res.vals))).max() < 1e-6
    assert np.all(np.diff(np.abs(fv)) <= 1e-12)                 # sorted by decreasing modulus
    assert np.allclose(spec[:, 0], ritz[:, 0], rtol=2e-7, atol=1e-12)   # (3E15.7) table vs full precision
    assert "3.00E-12" in out.stdout and "3.00E-02" in out.stdout and "nproj = 32" in out.stdout      # read from the case file, not hard-coded
    h.close()


def test_fortran_krylov_schur_restart_and_modes(tmp_path, spectre):
    """The whole reference loop from Fortran (host/krylov_host.f90): Krylov-Schur with schur_tgt = 2 (restarts through dgees /
    dtrsen and nsk_basis_gemm), spectra tables, eigenmode assembly (nsk_basis_gemv) and normalisation.  The converged leading
    pair is the reference's Spectre_Hd.dat row 1, the written mode satisfies M (Re + i Im) = mu (Re + i Im) through the Python
    binding of the same library, and the first restart's H equals the Python host's."""
    exe = os.path.join(ROOT, "host", "arnoldi_host")
    if not os.path.exists(exe):
        pytest.skip("host/arnoldi_host not built")
    from nekstab_amd import casefile, krylov, mesh, seed
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    qx, qy = seed.add_noise(case)
    pr = np.zeros((case.nel, 4, 4))
    cb = str(tmp_path / "case.bin")
    casefile.write_case_bin(cb, case, (qx, qy, pr))
    k = 48
    out = subprocess.run([exe, cb, str(k), str(tmp_path), "2", "d"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    print(out.stdout[-600:])
    assert "Schur restart 1" in out.stdout
    ritz = np.loadtxt(str(tmp_path / "ritz_full.txt"))
    fv = ritz[:, 0] + 1j * ritz[:, 1]
    ref = complex(*spectre["Hd"][0, :2])
    j = int(np.argmin(np.abs(fv - ref)))
    assert abs(fv[j] - ref) < 5e-7 and ritz[j, 2] < 1e-6, (fv[j], ref, ritz[j, 2])
    conv = np.loadtxt(str(tmp_path / "Spectre_NSd_conv.dat"))
    assert conv.shape[0] >= 2                                         # schur_tgt = 2 converged pairs
    assert abs(complex(conv[0, 0], abs(conv[0, 1])) - complex(*spectre["NSd_conv"][0, :2])) < 2e-6
    # the written mode: unit norm and an eigenvector of the map
    from nekstab_amd.settings import production_context
    h = production_context(case)
    n, m = case.lx1, case.lx1 - 2
    def load(name):
        raw = np.fromfile(str(tmp_path / name))
        nv = case.nel * n * n
        return raw[:nv].reshape(case.nel, n, n), raw[nv:2 * nv].reshape(case.nel, n, n), raw[2 * nv:].reshape(case.nel, m, m)
    re_, im_ = load("dRe00001.bin"), load("dIm00001.bin")
    vr, vi, fr, fi = h.alloc(4)
    h.upload(vr, *re_); h.upload(vi, *im_)
    assert abs(h.dot(vr, vr) + h.dot(vi, vi) - 1.0) < 1e-10          # core/eigensolvers.f:619-622
    h.matvec(fr, vr, 0); h.matvec(fi, vi, 0)
    mu = fv[j] if ritz[j, 1] > 0 else np.conj(fv[j])
    # file 00001 is the first converged row of the table: its eigenvalue is row `first` of ritz_full
    first = int(np.argmax(ritz[:, 2] < 1e-6))
    mu = fv[first]
    h.axpy(fr, -mu.real, vr); h.axpy(fr, mu.imag, vi)                # M Re - (mu_r Re - mu_i Im)
    h.axpy(fi, -mu.imag, vr); h.axpy(fi, -mu.real, vi)               # M Im - (mu_i Re + mu_r Im)
    resid = np.sqrt(h.dot(fr, fr) + h.dot(fi, fi))
    print("eigen-relation residual of the Fortran-written mode:", resid)
    assert resid < 1e-5
    # the same Krylov-Schur run from the Python host (nekstab_amd/krylov.py): same restarts, same converged pair
    v0 = h.alloc(1)[0]
    h.upload(v0, qx, qy, pr)
    res = krylov.krylov_schur(h, v0, k, schur_tgt=2)
    nrest = int(out.stdout.count("Schur restart"))
    lead = res.vals[np.argmin(np.abs(res.vals - ref))]
    print("restarts: Fortran %d, Python %d; leading pair: Fortran %s, Python %s" % (nrest, res.schur_cnt, fv[j], lead))
    assert abs(nrest - res.schur_cnt) <= 1
    assert abs(lead - fv[j]) < 1e-6
    h.close()


def test_reference_arnoldi_loop_drives_the_library_through_the_seam(tmp_path):
    """The boundary in the reference's own shape: host/_ref/ref_seam_driver links the REFERENCE'S
    core/krylov_decomposition.f -- compiled unchanged from /root/reference by `make -C host ref_seam` -- against
    host/ref_seam/krylov_subspace_hip.f90 (module krylov_subspace: type(krylov_vector) = a device handle; krylov_inner_product /
    norm / normalize / cmult / add2 / sub2 / zero / copy / matmul and matvec with the reference's argument lists).  Its
    arnoldi_factorization + update_hessenberg_matrix (modified Gram-Schmidt, two passes, one krylov_* call per operation) must
    produce the Hessenberg matrix of the Python host's loop (nsk_orth: batched two-pass classical Gram-Schmidt) on the same
    seed, at inner-solver tolerances tight enough that the comparison sees the seam and not the solvers.  This is a test of the
    BOUNDARY (stand-in SIZE / TOTAL declare nid, mstep, ifres only), not an oracle."""
    exe = os.path.join(ROOT, "host", "_ref", "ref_seam_driver")
    if not os.path.exists(exe):
        if shutil.which("flang") is None or not os.path.exists("/root/reference/core/krylov_decomposition.f"):
            pytest.skip("host/_ref/ref_seam_driver not prebuilt and the reference's source / flang are not here")
        subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "ref_seam"], check=True)
    from nekstab_amd import casefile, krylov, mesh, seed
    from nekstab_amd.capi import NekStabHip
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    qx, qy = seed.add_noise(case)
    pr = np.zeros((case.nel, 4, 4))
    tight = dict(tol_helm=1e-13, tol_pres=1e-9, min_pres_iter=2, nproj=8, max_helm_iter=200)
    cb = str(tmp_path / "case.bin")
    casefile.write_case_bin(cb, case, (qx, qy, pr), settings=tight)
    k = 12
    out = subprocess.run([exe, cb, str(k), str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "nsteps = 100" in out.stdout
    assert out.stdout.count("iteration current and total:") == k and out.stdout.count("ALPHA REORTH") == k * (k + 1) // 2   # the reference's own prints
    assert "live vectors before release: %d" % (k + 2) in out.stdout and "live vectors after release: 0" in out.stdout      # Q(1:k+1) + comb; f and wrk were finalised
    Hf = np.array(open(str(tmp_path / "HES_seam.txt")).read().split(), dtype=float)[: (k + 1) * k].reshape(k + 1, k)
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=tight["tol_helm"], tol_pres=tight["tol_pres"], tol_relative=1,
                   schwarz_layers=2, max_helm_iter=200, max_pres_iter=192, nproj=8)
    h.set_option("min_pres_iter", 2)
    Q = h.alloc(k + 1)
    h.upload(Q[0], qx, qy, pr)
    h.scal(Q[0], 1.0 / h.norm(Q[0]))
    H = np.zeros((k + 1, k))
    krylov.arnoldi_factorization(h, Q, H, 1, k, 0)
    err = np.abs(Hf - H).max() / np.abs(H).max()
    print("reference loop through the seam vs Python host: max |dH| / max |H| = %.2e" % err)
    assert err < 1e-10
    # krylov_matmul against the same combination through the Python binding
    c = h.alloc(1)[0]
    h.basis_gemv(Q[:3], np.array([0.6, -0.3, 0.2]), c)
    nrm = float([l for l in out.stdout.splitlines() if "|Q(1:3) y|" in l][0].split("=")[1])
    assert abs(nrm - h.norm(c)) < 1e-10
    h.close()


def test_fortran_host_checkpoint_restart_and_nek_mode_files(tmp_path):
    """Fortran-host parity with the Python host: host/nek_fld.f90 restates the reference's
    `arnoldi_checkpoint` (KRY<session>0.f<k+1>, HES<session><k>, Spectre_H/NS<op><k>.dat: core/eigensolvers.f:802-905), the restart
    from those files (:284-325) and `outpost` of the converged modes as Nek field files (:625-642), in the formats
    nekstab_amd/checkpoint.py / nekio.py use.  An interrupted-and-restarted Fortran run reproduces the uninterrupted one, the
    PYTHON host restarts from the Fortran host's files, and the written mode files are unit-norm eigenmodes."""
    exe = os.path.join(ROOT, "host", "arnoldi_host")
    if not os.path.exists(exe):
        if shutil.which("flang") is None:
            pytest.skip("flang not available and host/arnoldi_host not prebuilt")
        subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True)
    from nekstab_amd import casefile, checkpoint, krylov, mesh, nekio, seed
    from nekstab_amd.capi import NekStabHip
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)
    qx, qy = seed.add_noise(case)
    pr = np.zeros((case.nel, 4, 4))
    tight = dict(tol_helm=1e-13, tol_pres=1e-9, min_pres_iter=2, nproj=8, max_helm_iter=200)
    cb = str(tmp_path / "case.bin")
    casefile.write_case_bin(cb, case, (qx, qy, pr), settings=tight, eigen_tol=1.0, maxmodes=2)      # (eigen_tol 1: a 10-step run has "converged" modes to write)
    k = 10
    full, part = tmp_path / "full", tmp_path / "part"
    full.mkdir(); part.mkdir()
    out = subprocess.run([exe, cb, str(k), str(full), "0", "d", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for i in range(1, k + 2):
        assert (full / ("KRY1cyl0.f%05d" % i)).exists()
    for i in range(1, k + 1):
        assert (full / ("HES1cyl%04d" % i)).exists() and (full / ("Spectre_Hd%04d.dat" % i)).exists() and (full / ("Spectre_NSd%04d.dat" % i)).exists()
    Hf = np.array(open(str(full / "HES.txt")).read().split(), dtype=float)[: (k + 1) * k].reshape(k + 1, k)
    Hc = np.array(open(str(full / ("HES1cyl%04d" % k))).read().split(), dtype=float).reshape(k + 1, k)
    assert np.array_equal(Hf, Hc)
    # interrupted after 5 steps (a 5-step run with checkpoints), then restarted to k = 10 in the same directory
    out5 = subprocess.run([exe, cb, "5", str(part), "0", "d", "1"], capture_output=True, text=True, timeout=600)
    assert out5.returncode == 0, out5.stdout[-2000:] + out5.stderr[-2000:]
    outr = subprocess.run([exe, cb, str(k), str(part), "0", "d", "1", "5"], capture_output=True, text=True, timeout=600)
    assert outr.returncode == 0 and "restarted from the checkpoint of Arnoldi step 5" in outr.stdout, outr.stdout[-2000:] + outr.stderr[-2000:]
    Hr = np.array(open(str(part / "HES.txt")).read().split(), dtype=float)[: (k + 1) * k].reshape(k + 1, k)
    err = np.abs(Hr - Hf).max() / np.abs(Hf).max()
    print("Fortran run restarted at step 5 vs uninterrupted: max |dH| / max |H| = %.1e" % err)
    assert err < 1e-8
    # the Python host restarts from the Fortran host's files
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=tight["tol_helm"], tol_pres=tight["tol_pres"], tol_relative=1,
                   schwarz_layers=2, max_helm_iter=200, max_pres_iter=144, nproj=8)
    h.set_option("min_pres_iter", 2)
    Q, H, m1 = checkpoint.load_checkpoint(h, case, str(full), k, 5, session="1cyl")
    assert m1 == 6 and np.abs(H[:6, :5] - Hf[:6, :5]).max() == 0.0
    krylov.arnoldi_factorization(h, Q, H, m1, k, 0)
    errp = np.abs(H - Hf).max() / np.abs(Hf).max()
    print("Python host restarted from the Fortran host's checkpoint: max |dH| / max |H| = %.1e" % errp)
    assert errp < 1e-8
    # Krylov vector files: what the Fortran host wrote for vector k+1 is the Python host's vector k+1 (same sequence)
    f = nekio.read_fld(str(full / ("KRY1cyl0.f%05d" % (k + 1))))
    assert f.wdsize == 8 and f.nx == 6 and f.nel == case.nel and f.istep == h.nsteps + 1
    vx, vy, _ = h.download(Q[k])
    sc = max(np.abs(vx).max(), np.abs(vy).max())
    assert max(np.abs(f.u[0][:, 0] - vx).max(), np.abs(f.u[1][:, 0] - vy).max()) < 1e-7 * sc
    assert np.abs(f.x[0][:, 0] - case.x).max() == 0.0
    # mode files in Nek format: |Re|^2 + |Im|^2 = 1 in the bm1s norm (core/eigensolvers.f:619-622)
    re, im = nekio.read_fld(str(full / "dRe1cyl0.f00001")), nekio.read_fld(str(full / "dIm1cyl0.f00001"))
    a, b = h.alloc(2)
    checkpoint.fields_to_state(h, case, a, re)
    checkpoint.fields_to_state(h, case, b, im)
    assert abs(h.dot(a, a) + h.dot(b, b) - 1.0) < 1e-12
    h.close()
