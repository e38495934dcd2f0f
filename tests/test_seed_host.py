"""The seed path on the host (add_noise / mth_rand, core/utils.f:344-408, :457-469).

mth_rand = cos(1e3 sin(1e3 sin(r))) amplifies one ulp of a sine to 1e-2, so the seed is defined only up to the sine it is
computed with.  This build computes it with a CORRECTLY ROUNDED sine (nekstab_amd/crtrig.py on the host, csrc/nsk_crtrig.hpp on
the device: the same sequence of IEEE operations), which makes the seed the same vector on every machine:
  * crtrig against 700-bit mpmath: correctly rounded on every sampled argument up to 4e10;
  * the device header's constant tables = the host's;
  * the reference's expression transcribed for flang (host/seed_check.f90, libm's sin) = seed._mth_rand bit for bit wherever
    libm's sines are correctly rounded (glibc's sin is not on ~0.15 % of arguments): > 98 % of the nodes -- this pins the formula and
    the operation order; with another order or a fused multiply-add essentially no node would agree."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT


def test_crtrig_is_correctly_rounded():
    mp = pytest.importorskip("mpmath")
    from nekstab_amd import crtrig
    mp.mp.prec = 700
    assert abs(sum(mp.mpf(p) for p in crtrig._parts()) - mp.pi / 2) < mp.mpf(2) ** -160
    rng = np.random.default_rng(7)
    for scale in (1.0, 1.0e3, 6.5e7, 3.0e9, 4.0e10):
        x = np.concatenate([rng.uniform(-scale, scale, 1500), np.array([0.0, scale, -scale, 1e-300, np.pi / 2, np.pi, 355.0 / 113.0])])
        s, c = crtrig.sin_cr(x), crtrig.cos_cr(x)
        for xi, si, ci in zip(x, s, c):
            assert si == float(mp.sin(mp.mpf(float(xi)))) and ci == float(mp.cos(mp.mpf(float(xi)))), (scale, xi)


def test_device_header_holds_the_same_constants():
    from nekstab_amd import crtrig
    parts, fact = crtrig.device_header_constants()
    hdr = open(os.path.join(ROOT, "nekstab_amd", "csrc", "nsk_crtrig.hpp")).read()
    assert "PIO2[9] = {%s};" % parts in hdr
    assert "INVFACT[32][2] = {%s};" % fact in hdr
    assert float.hex(0.6366197723675814) == "0x1.45f306dc9c883p-1" and "0x1.45f306dc9c883p-1" in hdr


@pytest.mark.parametrize("ndim", [2, 3])
def test_flang_transcription_of_mth_rand_agrees_bit_for_bit_where_libm_is_correctly_rounded(case6, tmp_path, ndim):
    exe = os.path.join(ROOT, "host", "seed_check")
    if not os.path.exists(exe):
        if not shutil.which("flang"):
            pytest.skip("host/seed_check not built and no flang here")
        subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "seed_check"], check=True, stdout=subprocess.DEVNULL)
    from nekstab_amd import seed
    n = case6.lx1
    nel = 300
    x, y = case6.x[:nel], case6.y[:nel]
    ieg = np.arange(1, nel + 1).reshape(nel, 1, 1) * np.ones((1, n, n))
    idx = np.arange(1, n + 1, dtype=np.float64)
    ix, iy = idx[None, None, :] * np.ones((nel, n, 1)), idx[None, :, None] * np.ones((nel, 1, n))
    z = 0.37 * x - 0.21 * y                              # (any third coordinate: the 3-D branch of the formula)
    iz = ((np.arange(nel) % n) + 1).reshape(nel, 1, 1) * np.ones((1, n, n))
    xl = (x, y) if ndim == 2 else (x, y, z)
    mine = [seed._mth_rand(ix, iy, iz, ieg, xl, seed.FCOEFF[c]) for c in range(ndim)]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([x.size, ndim], dtype="<i4").tofile(f)
        for a in (ix, iy, iz, ieg):
            a.astype("<i4").tofile(f)
        for a in (x, y, z):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array(seed.FCOEFF, dtype="<f8").tofile(f)    # Fortran fcoeff(3, 3): column c = triple c (C order rows = Fortran columns)
    subprocess.run([exe, fin, fout], check=True)
    ref = np.fromfile(fout, dtype="<f8").reshape(ndim, -1)
    for c in range(ndim):
        a, b = mine[c].ravel(), ref[c]
        same = a == b
        print("ndim %d component %d: %d of %d nodes bit-identical (%.2f %%), largest difference elsewhere %.1e" % (ndim, c, same.sum(), a.size, 100.0 * same.mean(), np.abs(a - b).max()))
        assert same.mean() > 0.98
        # ... and with the HOST'S libm sine in the same expression the mirror reproduces the transcription on every node: the
        # differences above are libm's sines, nothing else
    import math
    vs, vc = np.vectorize(math.sin), np.vectorize(math.cos)
    from nekstab_amd import crtrig
    keep = (crtrig.sin_cr, crtrig.cos_cr)
    try:
        crtrig.sin_cr, crtrig.cos_cr = vs, vc
        libm = [seed._mth_rand(ix, iy, iz, ieg, xl, seed.FCOEFF[c]) for c in range(ndim)]
    finally:
        crtrig.sin_cr, crtrig.cos_cr = keep
    for c in range(ndim):
        assert np.array_equal(libm[c].ravel(), ref[c])
