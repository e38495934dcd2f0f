"""``outpost_ks`` (core/eigensolvers.f:508-721): spectra tables and converged eigenmodes in the
reference's on-disk formats, so nekStab's own post-processing (p_spec.py, wave_maker, ...) can read
them unchanged.

  Spectre_H<op>.dat        (3E15.7)  Re mu, Im mu, residual          :590
  Spectre_NS<op>.dat       (3E15.7)  Re/Im log(mu)/T, residual      :593-595
  Spectre_NS<op>_conv.dat  (2E15.7)  converged & outposted <= maxmodes   :601-604
  <op>Re<session>0.f0000i / <op>Im...  eigenmode i = Q y_i, |Re|^2+|Im|^2 = 1 (bm1s), time = i  :607-642
"""
from __future__ import annotations

import os

import numpy as np

from . import nekio
from .checkpoint import state_to_fields
from .krylov import assemble_mode, log_transform


def outpost_ks(be, res, case, outdir, *, evop="d", sampling_period=1.0, eigen_tol=1e-6, maxmodes=20,
               session="1cyl", wdsize=4):
    os.makedirs(outdir, exist_ok=True)
    k = res.H.shape[1]
    lam = log_transform(res.vals, sampling_period)
    nekio.write_spectre(os.path.join(outdir, f"Spectre_H{evop}.dat"), res.vals, res.residual)
    nekio.write_spectre(os.path.join(outdir, f"Spectre_NS{evop}.dat"), lam, res.residual)
    conv = []
    re, im = be.alloc(2)
    written = []
    for i in range(k):
        if res.residual[i] < eigen_tol and len(conv) < maxmodes:
            conv.append(lam[i])
            assemble_mode(be, res, i, re, im)
            idx = len(conv)
            for tag, v in (("Re", re), ("Im", im)):
                x, u, p1 = state_to_fields(be, case, v)          # pressure -> mesh 1 for output (map21); 2-D and 3-D
                f = os.path.join(outdir, "%s%s%s0.f%05d" % (evop, tag, session, idx))
                nekio.write_fld(f, x=x, u=u, p=p1, time=float(i + 1), istep=be.nsteps + 1, wdsize=wdsize)
                written.append(f)
    nekio.write_spectre(os.path.join(outdir, f"Spectre_NS{evop}_conv.dat"), np.array(conv))
    be.free([re, im])
    return written
