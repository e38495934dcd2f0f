"""nekStab's deterministic pseudo-noise seed, restated on the host
(``add_noise`` + ``mth_rand``, core/utils.f:344-408, :457-469): per-node value from
(i, j, k, global element id, coordinates), then dssum / multiplicity average and the
Dirichlet mask.  The device version is ``nsk_seed_noise``: the same operations in the same order, no fused
multiply-adds, and the correctly rounded sin / cos of ``crtrig`` on both sides -- the two agree BIT FOR BIT
(tests/test_kernels_gpu.py).  ``mth_rand`` amplifies one ulp of a sine to 1e-2, so a seed is only defined up to the sine
it was computed with: with a correctly rounded one it is the same vector on every machine.  host/seed_check.f90 is the
reference's formula transcribed for flang (libm's sin): it agrees with this file bit for bit on every node where libm's sines
are correctly rounded (> 98 % of the nodes; tests/test_seed_host.py)."""
from __future__ import annotations

import numpy as np

from . import crtrig

# fcoeff triples of add_noise (core/utils.f:372-381): qx, qy, qz
FCOEFF = ((3.0e4, -1.5e3, 0.5e5), (2.3e4, 2.3e3, -2.0e5), (2.0e4, 1.0e3, 1.0e5))


def _mth_rand(ix, iy, iz, ieg, xl, fc):
    """core/utils.f:457-469; ``xl`` = (x, y) or (x, y, z)."""
    sin, cos = crtrig.sin_cr, crtrig.cos_cr
    r = fc[0] * (ieg + xl[0] * sin(xl[1])) + fc[1] * ix * iy + fc[2] * ix
    if len(xl) == 3:                                   # IF3D branch (:463)
        r = fc[0] * (ieg + xl[2] * sin(r)) + fc[1] * iz * ix + fc[2] * iz
    r = 1.0e3 * sin(r)
    r = 1.0e3 * sin(r)
    return cos(r)


def add_noise(case):
    """Returns (qx, qy) shaped (nel, lx1, lx1), or (qx, qy, qz) shaped (nel, lx1, lx1, lx1) for hexahedra."""
    n = case.lx1
    nd = int(getattr(case, "ndim", 2))
    ieg = np.arange(1, case.nel + 1).reshape((case.nel,) + (1,) * nd)
    one = np.ones((1,) + (n,) * nd)
    idx = np.arange(1, n + 1, dtype=np.float64)
    if nd == 2:
        ix, iy, iz = idx[None, None, :] * one, idx[None, :, None] * one, None
        xl = (case.x, case.y)
    else:
        ix, iy, iz = idx[None, None, None, :] * one, idx[None, None, :, None] * one, idx[None, :, None, None] * one
        xl = (case.x, case.y, case.z)
    g = case.gid.ravel()
    vmult = 1.0 / np.bincount(g, minlength=case.nglob)[case.gid]      # Nek's VMULT: the INVERSE multiplicity, multiplied in (opcolv, dsavg's col2)

    def dssum(f):
        return np.bincount(g, weights=f.ravel(), minlength=case.nglob)[case.gid]

    out = []
    for c in range(nd):
        q = _mth_rand(ix, iy, iz, ieg, xl, FCOEFF[c])
        q = dssum(q) * vmult           # opdssum + opcolv(vmult)
        q = dssum(q * vmult)           # dsavg
        out.append(q * case.mask)      # bcdirvc
    return tuple(out)
