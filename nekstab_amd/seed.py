"""nekStab's deterministic pseudo-noise seed, restated on the host
(``add_noise`` + ``mth_rand``, core/utils.f:344-408, :457-469): per-node value from
(i, j, global element id, coordinates), then dssum / multiplicity average and the
Dirichlet mask. Used once per run; not on the device path."""
from __future__ import annotations

import numpy as np


def _mth_rand(ix, iy, ieg, x, y, fc):
    r = fc[0] * (ieg + x * np.sin(y)) + fc[1] * ix * iy + fc[2] * ix
    r = 1.0e3 * np.sin(r)
    r = 1.0e3 * np.sin(r)
    return np.cos(r)


def add_noise(case):
    """Returns (qx, qy) shaped (nel, lx1, lx1)."""
    n = case.lx1
    ix = np.arange(1, n + 1)[None, None, :] * np.ones((1, n, 1))
    iy = np.arange(1, n + 1)[None, :, None] * np.ones((1, 1, n))
    ieg = np.arange(1, case.nel + 1)[:, None, None]
    qx = _mth_rand(ix, iy, ieg, case.x, case.y, (3.0e4, -1.5e3, 0.5e5))
    qy = _mth_rand(ix, iy, ieg, case.x, case.y, (2.3e4, 2.3e3, -2.0e5))
    g = case.gid.ravel()
    mult = np.bincount(g, minlength=case.nglob)[case.gid]

    def dssum(f):
        return np.bincount(g, weights=f.ravel(), minlength=case.nglob)[case.gid]

    out = []
    for q in (qx, qy):
        q = dssum(q) / mult            # opdssum + opcolv(vmult)
        q = dssum(q / mult)            # dsavg
        out.append(q * case.mask)      # bcdirvc
    return out[0], out[1]
