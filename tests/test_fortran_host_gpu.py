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
    # same factorisation from the Python host
    from nekstab_amd.capi import NekStabHip
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1,
                   schwarz_layers=2, max_helm_iter=100, max_pres_iter=48, nproj=8)
    v0 = h.alloc(1)[0]
    h.upload(v0, qx, qy, pr)
    res = krylov.krylov_schur(h, v0, k, schur_tgt=0)
    # iterative inner solves => agreement to the solver tolerance, not bitwise
    assert np.abs(Hf - res.H).max() < 1e-7 * np.abs(res.H).max()
    fv = ritz[:, 0] + 1j * ritz[:, 1]
    assert np.abs(np.sort(np.abs(fv)) - np.sort(np.abs(res.vals))).max() < 1e-6
    assert np.all(np.diff(np.abs(fv)) <= 1e-12)                 # sorted by decreasing modulus
    assert np.allclose(spec[:, 0], ritz[:, 0], rtol=2e-7, atol=1e-12)   # (3E15.7) table vs full precision
    h.close()
