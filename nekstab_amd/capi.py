"""ctypes binding of libnekstab_hip.so (include/nekstab_hip.h).

This is the same thin layer a Fortran ``iso_c_binding`` interface block would be
(INTEGRATION.md); Python only carries pointers.  There is NO CPU fallback: if
the HIP library is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnekstab_hip.so")

NSK_DIRECT, NSK_ADJOINT, NSK_DIRECT_ADJOINT, NSK_NEWTON, NSK_FORCE_SENSITIVITY = 0, 1, 2, 3, 4

_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_longlong)
_ip = C.POINTER(C.c_int)


class NskCase(C.Structure):
    _fields_ = [
        ("ndim", C.c_int), ("nel", C.c_int), ("lx1", C.c_int), ("lxd", C.c_int),
        ("nglob", C.c_longlong),
        ("x", _dp), ("y", _dp), ("gid", _lp), ("mask", _dp), ("ub", _dp), ("vb", _dp),
        ("spng", _dp), ("vert", _lp), ("nvert", C.c_longlong),
        ("re", C.c_double), ("endtime", C.c_double), ("cfl", C.c_double),
        ("has_outflow", C.c_int), ("tol_helm", C.c_double), ("tol_pres", C.c_double),
        ("tol_relative", C.c_int), ("schwarz_layers", C.c_int),
        ("max_helm_iter", C.c_int), ("max_pres_iter", C.c_int), ("nproj", C.c_int),
        ("z", _dp), ("wb", _dp),
    ]


class NskStats(C.Structure):
    _fields_ = [("steps", C.c_longlong), ("helm_iters", C.c_longlong), ("pres_iters", C.c_longlong),
                ("unconverged", C.c_longlong), ("last_helm_res", C.c_double), ("last_pres_res", C.c_double),
                ("max_helm_iter", C.c_longlong), ("max_pres_iter", C.c_longlong),
                ("budget_helm", C.c_longlong), ("budget_pres", C.c_longlong),
                ("recaptures", C.c_longlong), ("retries", C.c_longlong),
                ("capped_solves", C.c_longlong), ("worst_cap_ratio", C.c_double),
                ("total_capped_solves", C.c_longlong), ("total_worst_cap_ratio", C.c_double),
                ("total_helm_iters", C.c_longlong), ("total_pres_iters", C.c_longlong), ("total_steps", C.c_longlong),
                ("recapture_seconds", C.c_double), ("total_pres_jsum", C.c_longlong), ("coarse_bytes_per_solve", C.c_double),
                ("step_budget_maps", C.c_longlong), ("step_budget_helm_mean", C.c_double), ("step_budget_pres_mean", C.c_double),
                ("tail_maps", C.c_longlong), ("zero_arrays", C.c_longlong)]


# every symbol include/nekstab_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "nsk_init": (C.c_int, [C.POINTER(NskCase), _vpp]),
    "nsk_finalize": (C.c_int, [_vp]),
    "nsk_init_local": (C.c_int, [C.POINTER(NskCase), C.POINTER(C.c_int), _vpp]),
    "nsk_local_info": (C.c_int, [_vp, _dp, _dp, _dp, _lp, _lp]),
    "nsk_local_rows": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "nsk_local_finish": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_longlong, C.c_longlong, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "nsk_shard_create_local": (C.c_int, [_vp, C.POINTER(C.c_int), _lp, C.c_int, C.c_int, _vpp]),
    "nsk_shard_share_stream": (C.c_int, [_vp, _vp]),
    "nsk_shard_elems": (C.c_int, [_vp, _lp]),
    "nsk_shard_halo_counts": (C.c_int, [_vp, _ip, _ip, _ip]),
    "nsk_last_error": (C.c_char_p, []),
    "nsk_get_info": (C.c_int, [_vp, _dp, C.POINTER(C.c_int), _lp, _lp, _lp]),
    "nsk_set_nsteps": (C.c_int, [_vp, C.c_int]),
    "nsk_set_tolerances": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    "nsk_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "nsk_vec_alloc": (C.c_int, [_vp, C.c_int, _vpp]),
    "nsk_vec_free": (C.c_int, [_vp, C.c_int, _vpp]),
    "nsk_vec_upload": (C.c_int, [_vp, _vp, _dp, _dp, _dp]),
    "nsk_vec_download": (C.c_int, [_vp, _vp, _dp, _dp, _dp]),
    "nsk_vec_upload_scalar": (C.c_int, [_vp, _vp, C.c_int, _dp]),
    "nsk_vec_download_scalar": (C.c_int, [_vp, _vp, C.c_int, _dp]),
    "nsk_vec_upload3": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp]),
    "nsk_vec_download3": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp]),
    "nsk_matvec": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "nsk_nonlinear_map": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "nsk_set_baseflow": (C.c_int, [_vp, _vp]),
    "nsk_set_orbit": (C.c_int, [_vp, _vp, C.c_double, _vp]),
    "nsk_dot": (C.c_int, [_vp, _vp, _vp, _dp]),
    "nsk_norm": (C.c_int, [_vp, _vp, _dp]),
    "nsk_scal": (C.c_int, [_vp, _vp, C.c_double]),
    "nsk_axpy": (C.c_int, [_vp, _vp, C.c_double, _vp]),
    "nsk_copy": (C.c_int, [_vp, _vp, _vp]),
    "nsk_zero": (C.c_int, [_vp, _vp]),
    "nsk_orth": (C.c_int, [_vp, _vp, _vpp, C.c_int, _dp, _dp]),
    "nsk_basis_gemm": (C.c_int, [_vp, _vpp, C.c_int, _dp, C.c_int]),
    "nsk_basis_gemv": (C.c_int, [_vp, _vpp, C.c_int, _dp, _dp, _vp, _vp]),
    "nsk_seed_noise": (C.c_int, [_vp, _vp]),
    "nsk_clone": (C.c_int, [_vp, _vpp]),
    "nsk_matvec_batch": (C.c_int, [_vpp, C.c_int, C.c_int, _vpp, _vpp]),
    "nsk_shard_create": (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int, C.c_int, _vpp]),
    "nsk_group_matvec": (C.c_int, [_vpp, C.c_int, C.c_int, _vpp, _vpp]),
    "nsk_group_nonlinear_map": (C.c_int, [_vpp, C.c_int, _vpp, _vpp, C.c_int]),
    "nsk_group_set_baseflow": (C.c_int, [_vpp, C.c_int, _vpp]),
    "nsk_group_set_orbit": (C.c_int, [_vpp, C.c_int, _vpp, C.c_double, _vpp]),
    "nsk_comm_init_host": (C.c_int, [_vp, _vp, _vp, _vp]),
    "nsk_shard_release_parent": (C.c_int, [_vp]),
    "nsk_comm_unique_id": (C.c_int, [_vp]),
    "nsk_comm_init_rccl": (C.c_int, [_vp, _vp]),
    "nsk_allreduce_host": (C.c_int, [_vp, _dp, C.c_int]),
    "nsk_group_test": (C.c_int, [_vpp, C.c_int, C.c_int, C.POINTER(_dp), C.POINTER(_dp)]),
    "nsk_local_dots": (C.c_int, [_vp, _vp, _vpp, C.c_int, _dp]),
    "nsk_project_out": (C.c_int, [_vp, _vp, _vpp, C.c_int, _dp]),
    "nsk_bench_kernel": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp]),
    "nsk_get_step_iters": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nsk_get_stats": (C.c_int, [_vp, C.POINTER(NskStats)]),
    "nsk_test_axhelm": (C.c_int, [_vp, _dp, C.c_double, C.c_double, _dp]),
    "nsk_test_dssum": (C.c_int, [_vp, _dp, _dp]),
    "nsk_test_opdiv": (C.c_int, [_vp, _dp, _dp, _dp]),
    "nsk_test_opgradt": (C.c_int, [_vp, _dp, _dp, _dp]),
    "nsk_test_convect": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "nsk_test_eapply": (C.c_int, [_vp, _dp, _dp]),
    "nsk_test_helm_solve": (C.c_int, [_vp, _dp, _dp, C.c_int, _dp, _dp, C.POINTER(C.c_int)]),
    "nsk_test_pres_solve": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_int)]),
    "nsk_test_op3": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, C.POINTER(C.c_int)]),
}


class NskError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnekstab_hip error {code}: {msg}")
        self.code = code


def load_library(path: str | None = None):
    """dlopen the C-ABI library and type every declared symbol (no compute)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.environ.get("NSK_LIB") or LIB_PATH      # NSK_LIB: diagnostic builds (stamps, experiments)
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the hot path)")
    try:
        # load order matters on this image: PyTorch-ROCm bundles its own HIP runtime; if this
        # library pulls in /opt/rocm's copy first, the process ends up with two runtimes and the
        # second one to initialise sees no device.  torch first => one shared runtime.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    diag = path is None and bool(os.environ.get("NSK_LIB"))      # A/B runs against older builds: tolerate missing exports
    for name, (res, args) in SYMBOLS.items():
        if diag and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = lib
    return lib


def _p(a):
    return a.ctypes.data_as(_dp)


class NekStabHip:
    """Device context for one case (``nsk_init`` ... ``nsk_finalize``)."""

    def __init__(self, case, vert, nvert, *, tol_helm=1e-9, tol_pres=1e-7, tol_relative=0,
                 schwarz_layers=2, max_helm_iter=80, max_pres_iter=40, nproj=0, local_own=None, ifbf2d=False):
        """``local_own`` (rank-local set-up, sharded.LocalParent): ``case`` is a rank's sub-mesh and local_own[e] = 1 marks the
        elements it owns -> nsk_init_local.  ``ifbf2d``: hexahedral run about a two-dimensional base flow -- the third base-flow
        component is forced to zero before the linearised solver is prepared (core/matvec.f:110-112, 'Forcing vz=0')."""
        self.lib = load_library()
        c = case
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.ndim = int(getattr(c, "ndim", 2))
        self._keep = dict(x=f64(c.x), y=f64(c.y), gid=np.ascontiguousarray(c.gid, dtype=np.int64),
                          mask=f64(c.mask), ub=f64(c.ub[0]), vb=f64(c.ub[1]), spng=f64(c.spng),
                          vert=np.ascontiguousarray(vert, dtype=np.int64))
        k = self._keep
        if self.ndim == 3:
            k["z"], k["wb"] = f64(c.z), (np.zeros_like(k["ub"]) if ifbf2d else f64(c.ub[2]))
        cs = NskCase(ndim=self.ndim, nel=c.nel, lx1=c.lx1, lxd=c.lxd, nglob=c.nglob,
                     x=_p(k["x"]), y=_p(k["y"]), gid=k["gid"].ctypes.data_as(_lp), mask=_p(k["mask"]),
                     ub=_p(k["ub"]), vb=_p(k["vb"]), spng=_p(k["spng"]),
                     vert=k["vert"].ctypes.data_as(_lp), nvert=int(nvert),
                     re=c.re, endtime=c.endtime, cfl=c.cfl, has_outflow=int(c.has_outflow),
                     tol_helm=tol_helm, tol_pres=tol_pres, tol_relative=tol_relative,
                     schwarz_layers=schwarz_layers, max_helm_iter=max_helm_iter,
                     max_pres_iter=max_pres_iter, nproj=nproj,
                     z=_p(k["z"]) if self.ndim == 3 else None, wb=_p(k["wb"]) if self.ndim == 3 else None)
        self.ctx = C.c_void_p()
        if local_own is None:
            self._chk(self.lib.nsk_init(C.byref(cs), C.byref(self.ctx)))
        else:
            own = np.ascontiguousarray(local_own, dtype=np.int32)
            self._chk(self.lib.nsk_init_local(C.byref(cs), own.ctypes.data_as(C.POINTER(C.c_int)), C.byref(self.ctx)))
        dt, ns = C.c_double(), C.c_int()
        a, b, d = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._chk(self.lib.nsk_get_info(self.ctx, C.byref(dt), C.byref(ns), C.byref(a), C.byref(b), C.byref(d)))
        self.dt, self.nsteps, self.nstate, self.nvel, self.npres = dt.value, ns.value, a.value, b.value, d.value
        self.nel, self.lx1, self.lx2 = c.nel, c.lx1, c.lx1 - 2

    def _chk(self, rc):
        if rc != 0:
            raise NskError(rc, self.lib.nsk_last_error().decode())

    def close(self):
        if self.ctx:
            for lane in getattr(self, "_lanes", [])[1:]:        # clones before the context they share their operators with
                self.lib.nsk_finalize(lane)
            self._lanes = []
            self.lib.nsk_finalize(self.ctx)
            self.ctx = C.c_void_p()

    # ---- vectors
    def alloc(self, n=1):
        arr = (C.c_void_p * n)()
        self._chk(self.lib.nsk_vec_alloc(self.ctx, n, arr))
        return [C.c_void_p(arr[i]) for i in range(n)]

    def free(self, vecs):
        arr = (C.c_void_p * len(vecs))(*[v.value for v in vecs])
        self._chk(self.lib.nsk_vec_free(self.ctx, len(vecs), arr))

    def set_nscal(self, nscal):
        """Carry ``nscal`` scalar fields (krylov_vector%theta) in every vector allocated from now on."""
        self.set_option("nscal", nscal)
        self.nscal = int(nscal)
        self.nstate = self.ndim * self.nvel + self.npres + self.nscal * self.nvel

    def upload_scalar(self, v, m, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        assert theta.size == self.nvel
        self._chk(self.lib.nsk_vec_upload_scalar(self.ctx, v, int(m), _p(theta)))

    def download_scalar(self, v, m):
        out = np.empty(self.nvel)
        self._chk(self.lib.nsk_vec_download_scalar(self.ctx, v, int(m), _p(out)))
        return out

    def upload3(self, v, vx, vy, vz, pr):
        vx, vy, vz, pr = (np.ascontiguousarray(a, dtype=np.float64) for a in (vx, vy, vz, pr))
        assert vx.size == self.nvel and vy.size == self.nvel and vz.size == self.nvel and pr.size == self.npres
        self._chk(self.lib.nsk_vec_upload3(self.ctx, v, _p(vx), _p(vy), _p(vz), _p(pr)))

    def download3(self, v):
        n, m = self.lx1, self.lx2
        vx = np.empty((self.nel, n, n, n)); vy = np.empty_like(vx); vz = np.empty_like(vx); pr = np.empty((self.nel, m, m, m))
        self._chk(self.lib.nsk_vec_download3(self.ctx, v, _p(vx), _p(vy), _p(vz), _p(pr)))
        return vx, vy, vz, pr

    def upload(self, v, vx, vy, pr):
        vx, vy, pr = (np.ascontiguousarray(a, dtype=np.float64) for a in (vx, vy, pr))
        assert vx.size == self.nvel and vy.size == self.nvel and pr.size == self.npres
        self._chk(self.lib.nsk_vec_upload(self.ctx, v, _p(vx), _p(vy), _p(pr)))

    def download(self, v):
        n, m = self.lx1, self.lx2
        vx = np.empty((self.nel, n, n)); vy = np.empty((self.nel, n, n)); pr = np.empty((self.nel, m, m))
        self._chk(self.lib.nsk_vec_download(self.ctx, v, _p(vx), _p(vy), _p(pr)))
        return vx, vy, pr

    # ---- lanes (nsk_clone / nsk_matvec_batch): independent maps in flight at once
    def add_lane(self):
        """A further lane of this context (own stream and state, shared operators).  Returns its index; lane 0 is the context."""
        if not hasattr(self, "_lanes"):
            self._lanes = [self.ctx]
        out = C.c_void_p()
        self._chk(self.lib.nsk_clone(self.ctx, C.byref(out)))
        self._lanes.append(out)
        return len(self._lanes) - 1

    def matvec_batch(self, fs, qs, mode=NSK_DIRECT):
        """f[k] = map(q[k]) on lane k, k = 0 .. len(qs) - 1, all maps in flight at once."""
        b = len(qs)
        lanes = getattr(self, "_lanes", [self.ctx])
        while len(lanes) < b:
            self.add_lane()
            lanes = self._lanes
        la = (C.c_void_p * b)(*[l.value for l in lanes[:b]])
        fa = (C.c_void_p * b)(*[v.value for v in fs])
        qa = (C.c_void_p * b)(*[v.value for v in qs])
        self._chk(self.lib.nsk_matvec_batch(la, b, mode, fa, qa))

    def lane_stats(self, k):
        st = NskStats()
        self._chk(self.lib.nsk_get_stats(self._lanes[k], C.byref(st)))
        return {f: getattr(st, f) for f, _ in NskStats._fields_}

    # ---- operator + vector algebra
    def matvec(self, f, q, mode=NSK_DIRECT):
        self._chk(self.lib.nsk_matvec(self.ctx, mode, f, q))

    def nonlinear_map(self, f, q, subtract_q=False):
        self._chk(self.lib.nsk_nonlinear_map(self.ctx, f, q, int(subtract_q)))

    def set_baseflow(self, q):
        self._chk(self.lib.nsk_set_baseflow(self.ctx, q))
        dt, ns = C.c_double(), C.c_int()
        a, b, d = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._chk(self.lib.nsk_get_info(self.ctx, C.byref(dt), C.byref(ns), C.byref(a), C.byref(b), C.byref(d)))
        self.dt, self.nsteps = dt.value, ns.value

    def set_orbit(self, q0, spng_str=0.0, end=None):
        self._chk(self.lib.nsk_set_orbit(self.ctx, q0, float(spng_str), end))
        dt, ns = C.c_double(), C.c_int()
        a, b, d = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._chk(self.lib.nsk_get_info(self.ctx, C.byref(dt), C.byref(ns), C.byref(a), C.byref(b), C.byref(d)))
        self.dt, self.nsteps = dt.value, ns.value

    def dot(self, p, q):
        a = C.c_double()
        self._chk(self.lib.nsk_dot(self.ctx, p, q, C.byref(a)))
        return a.value

    def norm(self, p):
        a = C.c_double()
        self._chk(self.lib.nsk_norm(self.ctx, p, C.byref(a)))
        return a.value

    def scal(self, p, a):
        self._chk(self.lib.nsk_scal(self.ctx, p, a))

    def axpy(self, p, a, q):
        self._chk(self.lib.nsk_axpy(self.ctx, p, a, q))

    def copy(self, dst, src):
        self._chk(self.lib.nsk_copy(self.ctx, dst, src))

    def zero(self, p):
        self._chk(self.lib.nsk_zero(self.ctx, p))

    def orth(self, f, Q):
        j = len(Q)
        arr = (C.c_void_p * max(j, 1))(*[v.value for v in Q])
        h = np.zeros(max(j, 1))
        beta = C.c_double()
        self._chk(self.lib.nsk_orth(self.ctx, f, arr, j, _p(h), C.byref(beta)))
        return h[:j].copy(), beta.value

    def basis_gemm(self, Q, Z):
        k = len(Q)
        Zc = np.asfortranarray(Z, dtype=np.float64)       # column-major, ldz = k
        arr = (C.c_void_p * k)(*[v.value for v in Q])
        self._chk(self.lib.nsk_basis_gemm(self.ctx, arr, k, Zc.ctypes.data_as(_dp), Zc.shape[0]))

    def basis_gemv(self, Q, y, re, im=None):
        k = len(Q)
        yr = np.ascontiguousarray(np.real(y), dtype=np.float64)
        yi = np.ascontiguousarray(np.imag(y), dtype=np.float64)
        arr = (C.c_void_p * k)(*[v.value for v in Q])
        self._chk(self.lib.nsk_basis_gemv(self.ctx, arr, k, _p(yr), _p(yi) if im is not None else None, re, im))

    def seed_noise(self, v):
        """add_noise (core/utils.f:344-408) on the device."""
        self._chk(self.lib.nsk_seed_noise(self.ctx, v))

    # Lanes (nsk_clone) hold their solver settings BY VALUE: a setter called after lanes exist acts on every lane, so that
    # matvec_batch / band_arnoldi never apply different operators on different lanes.
    def _all_lanes(self):
        return getattr(self, "_lanes", None) or [self.ctx]

    def set_nsteps(self, n):
        for lane in self._all_lanes():
            self._chk(self.lib.nsk_set_nsteps(lane, n))
        self.nsteps = n

    def set_tolerances(self, th, tp, relative=0):
        for lane in self._all_lanes():
            self._chk(self.lib.nsk_set_tolerances(lane, th, tp, relative))

    def set_option(self, name, value):
        for lane in self._all_lanes():
            self._chk(self.lib.nsk_set_option(lane, name.encode(), float(value)))

    def bench_kernel(self, name, reps=200):
        us = C.c_double()
        self._chk(self.lib.nsk_bench_kernel(self.ctx, name.encode(), reps, C.byref(us)))
        return {"kernel": name, "avg_us": us.value, "reps": reps}

    def step_iters(self):
        """(helm, pres): iteration counts of every time step of this context's last map"""
        n = C.c_int(0)
        self._chk(self.lib.nsk_get_step_iters(self.ctx, 0, None, None, C.byref(n)))
        hh = np.zeros(max(n.value, 1), dtype=np.int32); pp = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self.lib.nsk_get_step_iters(self.ctx, n.value, hh.ctypes.data_as(C.POINTER(C.c_int)), pp.ctypes.data_as(C.POINTER(C.c_int)), C.byref(n)))
        return hh[:n.value], pp[:n.value]

    def stats(self):
        s = NskStats()
        self._chk(self.lib.nsk_get_stats(self.ctx, C.byref(s)))
        return {f: getattr(s, f) for f, _ in NskStats._fields_}

    # ---- kernel-level test hooks
    def t_axhelm(self, u, h1, h2):
        u = np.ascontiguousarray(u); out = np.empty_like(u)
        self._chk(self.lib.nsk_test_axhelm(self.ctx, _p(u), h1, h2, _p(out)))
        return out

    def t_dssum(self, u):
        u = np.ascontiguousarray(u); out = np.empty_like(u)
        self._chk(self.lib.nsk_test_dssum(self.ctx, _p(u), _p(out)))
        return out

    def t_opdiv(self, u, v):
        u = np.ascontiguousarray(u); v = np.ascontiguousarray(v)
        out = np.empty((self.nel, self.lx2, self.lx2))
        self._chk(self.lib.nsk_test_opdiv(self.ctx, _p(u), _p(v), _p(out)))
        return out

    def t_opgradt(self, p):
        p = np.ascontiguousarray(p)
        ox = np.empty((self.nel, self.lx1, self.lx1)); oy = np.empty_like(ox)
        self._chk(self.lib.nsk_test_opgradt(self.ctx, _p(p), _p(ox), _p(oy)))
        return ox, oy

    def t_convect(self, u, v, adjoint=False):
        u = np.ascontiguousarray(u); v = np.ascontiguousarray(v)
        ox = np.empty_like(u); oy = np.empty_like(u)
        self._chk(self.lib.nsk_test_convect(self.ctx, int(adjoint), _p(u), _p(v), _p(ox), _p(oy)))
        return ox, oy

    def t_eapply(self, p):
        p = np.ascontiguousarray(p); out = np.empty_like(p)
        self._chk(self.lib.nsk_test_eapply(self.ctx, _p(p), _p(out)))
        return out

    def t_helm_solve(self, rx, ry, order=3):
        rx = np.ascontiguousarray(rx); ry = np.ascontiguousarray(ry)
        ox = np.empty_like(rx); oy = np.empty_like(rx); it = C.c_int()
        self._chk(self.lib.nsk_test_helm_solve(self.ctx, _p(rx), _p(ry), order, _p(ox), _p(oy), C.byref(it)))
        return ox, oy, it.value

    def t_op3(self, which, inp, a=0):
        """3-D element operators on packed arrays (nsk_test_op3)."""
        n, m = self.lx1, self.lx2
        inp = np.ascontiguousarray(inp, dtype=np.float64)
        out = np.empty((self.nel, m, m, m)) if which in (1, 6, 7) else np.empty((3, self.nel, n, n, n))      # 3, 8: convection
        it = C.c_int()
        self._chk(self.lib.nsk_test_op3(self.ctx, which, _p(inp), _p(out), int(a), C.byref(it)))
        return (out, it.value) if which == 5 else out

    def t_pres_solve(self, g):
        g = np.ascontiguousarray(g); out = np.empty_like(g); it = C.c_int()
        self._chk(self.lib.nsk_test_pres_solve(self.ctx, _p(g), _p(out), C.byref(it)))
        return out, it.value
