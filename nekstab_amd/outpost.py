"""``outpost_ks`` (core/eigensolvers.f:508-721): spectra tables and converged eigenmodes in the
reference's on-disk formats, so nekStab's own post-processing (p_spec.py, wave_maker, ...) can read
them unchanged.

  Spectre_H<op>.dat        (3E15.7)  Re mu, Im mu, residual          :590
  Spectre_NS<op>.dat       (3E15.7)  Re/Im log(mu)/T, residual      :593-595
  Spectre_NS<op>_conv.dat  (2E15.7)  converged & outposted <= maxmodes   :601-604
  <op>Re<session>0.f0000i / <op>Im...  eigenmode i = Q y_i, |Re|^2+|Im|^2 = 1 (bm1s), time = i  :607-642
  Spectre_<op>.info        key=value manifest of the run (mesh, uparam, solver, eigensolver)   :664-717
"""
from __future__ import annotations

import os

import numpy as np

from . import nekio
from .checkpoint import state_to_fields
from .krylov import assemble_mode, log_transform


def _fortran_e(x, width, digits):
    """Fortran ``Ew.d`` edit descriptor (0.1234567E+01): python's %E normalises to 1.234567E+00."""
    if x == 0.0:
        m, e = 0.0, 0
    else:
        e = int(np.floor(np.log10(abs(x)))) + 1
        m = x / 10.0 ** e
        if abs(round(m, digits)) >= 1.0:              # 0.99999996 -> 1.0000000: renormalise like the Fortran runtime
            m /= 10.0
            e += 1
    return ("%.*fE%+03d" % (digits, m, e)).rjust(width)


def write_info(path, be, case, res, *, evop, sampling_period, eigen_tol, schur_tgt, schur_del, outposted, uparam=None,
               tol_pres=None, tol_vel=None, nranks=1, ctarg=None, k_dim=None):
    """``Spectre_<op>.info`` as outpost_ks writes it (core/eigensolvers.f:664-717): formats (A,I16), (A,F16.4),
    (A,F16.12), (A,E15.7), (A,E13.4); same keys in the same order.  Version strings name this build."""
    up = list(uparam) if uparam is not None else [0.0] * 10
    up += [0.0] * (10 - len(up))
    nd = int(getattr(case, "ndim", 2))
    L = ["Nek5000 version:" + "none (nekstab_amd: MI355X-native time stepper)", "nekStab version:" + "nekstab_amd",
         "[mesh]",
         "lx1=             %16d" % case.lx1, "polyOrder N=     %16d" % (case.lx1 - 1), "tot elemts=      %16d" % case.nel,
         "tot points=      %16d" % (case.nel * case.lx1 ** nd), "MPI ranks=       %16d" % nranks, "e/rank=          %16d" % (case.nel // nranks),
         "[userParams]"]
    L += ["uparam%02d=        %16.12f" % (i + 1, up[i]) for i in range(10)]
    L += ["[solver]",
          "ctarg=           %16.4f" % (case.cfl if ctarg is None else ctarg), "nsteps=          %16d" % be.nsteps,
          "dt=              " + _fortran_e(be.dt, 15, 7), "Re=              %16.4f" % case.re,
          "residualTol PRE= " + _fortran_e(0.0 if tol_pres is None else tol_pres, 13, 4),
          "residualTol VEL= " + _fortran_e(0.0 if tol_vel is None else tol_vel, 13, 4),
          "[eigensolver]",
          "sampling period =%16.12f" % sampling_period, "k_dim=           %16d" % (res.H.shape[1] if k_dim is None else k_dim),
          "eigentol=        " + _fortran_e(eigen_tol, 13, 4), "schur_target=    %16d" % schur_tgt, "schur_del=       %16.4f" % schur_del,
          "schur iterations=%16d" % res.schur_cnt, "outposted=       %16d" % outposted]
    with open(path, "w") as fh:
        fh.write("\n".join(L) + "\n")


def read_info(path):
    """key -> string value of a ``Spectre_<op>.info`` manifest (ours or the reference's)."""
    out = {}
    for line in open(path):
        line = line.rstrip("\n")
        if "=" in line and not line.startswith("["):
            k, v = line.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def outpost_ks(be, res, case, outdir, *, evop="d", sampling_period=1.0, eigen_tol=1e-6, maxmodes=20,
               session="1cyl", wdsize=4, schur_tgt=0, schur_del=0.10, uparam=None, tol_pres=None, tol_vel=None, nranks=1):
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
    write_info(os.path.join(outdir, f"Spectre_{evop}.info"), be, case, res, evop=evop, sampling_period=sampling_period,
               eigen_tol=eigen_tol, schur_tgt=schur_tgt, schur_del=schur_del, outposted=len(conv), uparam=uparam,
               tol_pres=tol_pres, tol_vel=tol_vel, nranks=nranks)
    be.free([re, im])
    return written
