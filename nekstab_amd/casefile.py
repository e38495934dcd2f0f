"""Binary case file for the Fortran host (host/arnoldi_host.f90): the arrays of ``nsk_case`` plus a
seed vector, little-endian, in the order the Fortran program reads them with stream access."""
import numpy as np


def write_case_bin(path, case, seed_state, settings=None, eigen_tol=1e-6, schur_del=0.10, maxmodes=20):
    """``settings``: inner-solver settings (default: nekstab_amd.settings.PRODUCTION + PRODUCTION_OPTIONS, what bench.py times
    and the spectrum tests pin); the eigensolver parameters are nekStab's defaults (core/usr_extra.f:9-29)."""
    from .settings import PRODUCTION, PRODUCTION_OPTIONS
    st = dict(PRODUCTION, **PRODUCTION_OPTIONS)
    st.update(settings or {})
    vx, vy, pr = seed_state
    with open(path, "wb") as f:
        np.array([case.ndim, case.nel, case.lx1, case.lxd, case.meta["nvert"], int(case.has_outflow), 0, 0],
                 dtype="<i4").tofile(f)
        np.array([case.nglob], dtype="<i8").tofile(f)
        np.array([case.re, case.endtime, case.cfl], dtype="<f8").tofile(f)
        for a in (case.x, case.y):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.ascontiguousarray(case.gid, dtype="<i8").tofile(f)
        for a in (case.mask, case.ub[0], case.ub[1], case.spng):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.ascontiguousarray(case.meta["vert"], dtype="<i8").tofile(f)
        for a in (vx, vy, pr):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array([st["tol_helm"], st["tol_pres"], st["min_pres_iter"], st["nproj"], eigen_tol, schur_del, maxmodes, st["max_helm_iter"]],
                 dtype="<f8").tofile(f)
