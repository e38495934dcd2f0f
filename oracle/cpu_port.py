"""CPU PORT OF THE HOT PATH -- TEST / BASELINE INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Host side of ``oracle/cpu_step.c`` (C + OpenMP): builds, from the numpy oracle's operators (``oracle/linns.py``,
which is pinned on the reference's golden data), the arrays the C time stepper needs -- geometry factors, the CSR of
co-located nodes for dssum, base-flow constants on the dealiasing mesh, Jacobi diagonals, the restricted additive
Schwarz patch inverses and the vertex coarse inverse of the pressure preconditioner (same construction as
``nekstab_amd/csrc/nsk.hip: build()``, from the oracle's assembled E instead of device probes) -- and drives whole
matvecs ``f = Phi_T q`` (core/matvec.f:163-243) and Arnoldi steps (core/krylov_decomposition.f:73-202) on the host
cores.  What bench.py times is :func:`CpuPort.arnoldi_steps`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libcpu_step.so")
SRC = os.path.join(HERE, "cpu_step.c")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


class CpuCase(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("nel", "N", "M", "ND", "nvert", "PS", "max_helm", "max_pres", "min_pres",
                                        "tol_relative", "helm_guess", "nproj")] +
                [(n, _dp) for n in ("D", "J12", "D12", "Jd", "Dd", "hat", "g1", "g2", "g4", "bm1", "mask", "minv", "binv",
                                    "spng", "dinv", "w2rx", "w2sx", "w2ry", "w2sy", "cUr", "cUs", "GUx", "GUy", "GVx", "GVy")] +
                [("gs_off", _ip), ("gs_idx", _ip), ("p_idx", _ip), ("p_inv", _fp), ("Aci", _fp),
                 ("evert", _ip), ("v_off", _ip), ("v_ent", _ip)] +
                [(n, C.c_double) for n in ("nu", "dt", "vol", "tol_helm", "tol_pres", "early_pres_mul")])


class CpuStats(C.Structure):
    _fields_ = [("steps", C.c_longlong), ("helm_iters", C.c_longlong), ("pres_iters", C.c_longlong),
                ("unconverged", C.c_longlong), ("last_helm_res", C.c_double), ("last_pres_res", C.c_double),
                ("last_pres_iters", C.c_longlong)]


CFLAGS = ["-O3", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared"]


def build_library(force=False):
    """gcc -O3 -fopenmp -> oracle/_build/libcpu_step.so (git-ignored, travels with the snapshot).  The cache is keyed on a
    hash of the source and the flags (not on mtimes: the snapshot that travels to the GPU box resets them)."""
    import hashlib
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(SRC, "rb") as fh:
        key = hashlib.sha256(fh.read() + " ".join(CFLAGS).encode()).hexdigest()
    stamp = LIB + ".srchash"
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or not os.path.exists(LIB) or have != key:
        subprocess.run(["gcc"] + CFLAGS + ["-o", LIB, SRC, "-lm"], check=True)
        with open(stamp, "w") as fh:
            fh.write(key)
    return LIB


def _load():
    lib = C.CDLL(build_library())
    lib.cpu_matvec.restype = C.c_int
    lib.cpu_matvec.argtypes = [C.POINTER(CpuCase), _dp, _dp, C.c_int, C.POINTER(CpuStats)]
    lib.cpu_set_threads.argtypes = [C.c_int]
    lib.cpu_max_threads.restype = C.c_int
    lib.cpu_proj_reset.restype = None
    lib.cpu_regions.restype = C.c_longlong
    lib.cpu_regions_reset.restype = None
    lib.cpu_empty_region_us.restype = C.c_double
    lib.cpu_empty_region_us.argtypes = [C.c_int, C.c_int]
    return lib


class CpuPort:
    def __init__(self, o, vert, nvert, *, tol_helm=1e-9, tol_pres=1e-7, tol_relative=0, min_pres=0, layers=2,
                 max_helm=120, max_pres=48, early_pres_mul=1e-2, helm_guess=1, nproj=0):
        """``o``: a LinNS2D oracle with its pressure matrix built; ``vert``: (nel,4) lexicographic vertex ids."""
        assert o.has_outflow, "the C port covers domains with an outflow boundary (configs 1 and 2)"
        self.lib = _load()
        self.o = o
        t0 = time.perf_counter()
        n, m, nel, nd = o.n, o.m, o.nel, o.lxd
        NN, MM = n * n, m * m
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        k = {}
        k["D"], k["J12"], k["D12"], k["Jd"], k["Dd"] = f64(o.D), f64(o.J12), f64(o.D12), f64(o.Jd), f64(o.Dd)
        hm, hp = 0.5 * (1 - o.z2), 0.5 * (1 + o.z2)
        hat = np.stack([np.outer(hm, hm), np.outer(hm, hp), np.outer(hp, hm), np.outer(hp, hp)]).reshape(4, MM)
        k["hat"] = f64(hat)
        gmask = o.gmask[o.gid]
        k["g1"], k["g2"], k["g4"], k["bm1"], k["mask"], k["spng"] = f64(o.g1), f64(o.g2), f64(o.g4), f64(o.bm1), f64(gmask), f64(o.spng)
        k["minv"] = f64(1.0 / o.mult)
        k["binv"] = f64(gmask * o.binvm1)
        # Jacobi diagonal of H for the three BDF orders (as nsk.hip build())
        D = o.D
        dA = np.einsum("ki,ejk->eji", D * D, o.g1) + np.einsum("kj,eki->eji", D * D, o.g2) + \
            2.0 * np.einsum("i,j,eji->eji", np.diag(D), np.diag(D), o.g4)
        dAs, bs = o.dssum(dA), o.dssum(o.bm1)
        k["dinv"] = f64(np.stack([gmask / (o.nu * dAs + b0 / o.dt * bs) for b0 in (1.0, 1.5, 11.0 / 6.0)]))
        k["w2rx"], k["w2sx"], k["w2ry"], k["w2sy"] = (f64(a * o.W2) for a in (o.rxm2, o.sxm2, o.rym2, o.sym2))
        U, V = o.ub
        Uf, Vf = o.Jd @ U @ o.Jd.T, o.Jd @ V @ o.Jd.T
        Ur, Us, Vr, Vs = Uf @ o.Dd.T, o.Dd @ Uf, Vf @ o.Dd.T, o.Dd @ Vf
        k["cUr"], k["cUs"] = f64(o.rxd * Uf + o.ryd * Vf), f64(o.sxd * Uf + o.syd * Vf)
        k["GUx"], k["GUy"] = f64(o.rxd * Ur + o.sxd * Us), f64(o.ryd * Ur + o.syd * Us)
        k["GVx"], k["GVy"] = f64(o.rxd * Vr + o.sxd * Vs), f64(o.ryd * Vr + o.syd * Vs)
        # dssum as a gather: CSR of co-located local nodes, ascending
        g = o.gid.ravel()
        order = np.argsort(g, kind="stable")
        cnt = np.bincount(g, minlength=o.nglob)
        start = np.concatenate([[0], np.cumsum(cnt)])
        off = np.concatenate([[0], np.cumsum(cnt[g])])
        idx = np.empty(off[-1], dtype=np.int64)
        pos = off[:-1]
        maxc = int(cnt.max())
        for r in range(maxc):
            sel = cnt[g] > r
            idx[pos[sel] + r] = order[start[g[sel]] + r]
        k["gs_off"], k["gs_idx"] = i32(off), i32(idx)
        # pressure preconditioner from the assembled E
        E = o._Emat.tocsr()
        vert = np.asarray(vert, dtype=np.int64).reshape(nel, 4)
        owner = [[] for _ in range(o.nglob)]
        ug = [np.unique(o.gid[e]) for e in range(nel)]
        for e in range(nel):
            for gg in ug[e]:
                owner[gg].append(e)
        L = max(1, min(layers, min(4, m)))
        PS = (((m + 2 * L) * (m + 2 * L) + 3) // 4) * 4
        p_idx = np.full((nel, PS), -1, dtype=np.int32)
        p_inv = np.zeros((nel, MM, PS), dtype=np.float32)
        colmap = np.full(o.npr, -1, dtype=np.int64)
        for e in range(nel):
            dof = list(range(e * MM, (e + 1) * MM))
            ge = ug[e]
            nbs = sorted({f for gg in ge for f in owner[gg] if f != e})
            for f in nbs:
                jj, ii = np.where(np.isin(o.gid[f], ge))
                jmin, jmax, imin, imax = jj.min(), jj.max(), ii.min(), ii.max()
                b0, b1, a0, a1 = 0, m, 0, m
                if jmin == jmax:
                    if jmin == 0: b1 = L
                    elif jmin == n - 1: b0 = m - L
                if imin == imax:
                    if imin == 0: a1 = L
                    elif imin == n - 1: a0 = m - L
                if (b0, b1, a0, a1) == (0, m, 0, m):
                    continue
                dof += [f * MM + b * m + a for b in range(b0, b1) for a in range(a0, a1)]
            dof = np.array(dof[:PS])
            npd = len(dof)
            colmap[dof] = np.arange(npd)
            sub = E[dof]
            rows = np.repeat(np.arange(npd), np.diff(sub.indptr))
            mc = colmap[sub.indices]
            keep = mc >= 0
            A = np.zeros((npd, npd))
            A[rows[keep], mc[keep]] = sub.data[keep]
            colmap[dof] = -1
            p_idx[e, :npd] = dof
            p_inv[e, :, :npd] = np.linalg.inv(A)[:MM, :]
        k["p_idx"], k["p_inv"] = p_idx, np.ascontiguousarray(p_inv)
        rows = (np.arange(nel)[:, None, None] * MM + np.arange(MM)[None, None, :]).repeat(4, 1).ravel()
        cols = vert[:, :, None].repeat(MM, 2).ravel()
        R = sp.coo_matrix((np.tile(hat, (nel, 1, 1)).ravel(), (rows, cols)), shape=(o.npr, int(nvert))).tocsr()
        Ac = (R.T @ E @ R).toarray()
        k["Aci"] = np.ascontiguousarray(np.linalg.inv(Ac), dtype=np.float32)
        k["evert"] = i32(vert)
        vcnt = np.bincount(vert.ravel(), minlength=int(nvert))
        k["v_off"] = i32(np.concatenate([[0], np.cumsum(vcnt)]))
        k["v_ent"] = i32(np.argsort(vert.ravel(), kind="stable"))
        self._keep = k
        ptr = {"gs_off": _ip, "gs_idx": _ip, "p_idx": _ip, "evert": _ip, "v_off": _ip, "v_ent": _ip, "p_inv": _fp, "Aci": _fp}
        args = {name: k[name].ctypes.data_as(ptr.get(name, _dp)) for name in k}
        self.case = CpuCase(nel=nel, N=n, M=m, ND=nd, nvert=int(nvert), PS=PS, max_helm=max_helm, max_pres=max_pres,
                            min_pres=min_pres, tol_relative=tol_relative, helm_guess=helm_guess, nproj=int(nproj),
                            nu=o.nu, dt=o.dt, vol=float(o.volvm1), tol_helm=tol_helm, tol_pres=tol_pres,
                            early_pres_mul=early_pres_mul, **args)
        self.nsteps = o.nsteps
        self.nvel, self.npres = nel * NN, nel * MM
        self.shape_v, self.shape_p = (nel, n, n), (nel, m, m)
        self.setup_s = time.perf_counter() - t0
        self.stats = None
        self.lib.cpu_proj_reset()

    def proj_reset(self):
        """empty the pressure projection space (it is kept from matvec to matvec, as the device's)"""
        self.lib.cpu_proj_reset()

    def regions(self, reset=False):
        """OpenMP parallel regions opened since the last reset"""
        n = int(self.lib.cpu_regions())
        if reset:
            self.lib.cpu_regions_reset()
        return n

    def empty_region_us(self, n=4096, reps=2000):
        """microseconds per empty parallel-for region at the current thread count"""
        return float(self.lib.cpu_empty_region_us(int(n), int(reps)))

    def set_threads(self, nthreads):
        self.lib.cpu_set_threads(int(nthreads))

    def max_threads(self):
        return int(self.lib.cpu_max_threads())

    def matvec(self, q, nsteps=None):
        """f = Phi_T q, direct map; q, f = (u, v, p) arrays shaped like the oracle's."""
        qq = np.concatenate([np.ravel(a) for a in q]).astype(np.float64)
        f = np.empty_like(qq)
        st = CpuStats()
        rc = self.lib.cpu_matvec(C.byref(self.case), qq.ctypes.data_as(_dp), f.ctypes.data_as(_dp), int(nsteps or self.nsteps), C.byref(st))
        if rc < 0:
            raise RuntimeError("cpu_matvec: bad arguments")
        self.stats = {n: getattr(st, n) for n, _ in CpuStats._fields_}
        nv = self.nvel
        return f[:nv].reshape(self.shape_v), f[nv:2 * nv].reshape(self.shape_v), f[2 * nv:].reshape(self.shape_p)

    def arnoldi_steps(self, q0, k, H=None):
        """k Arnoldi steps (matvec + two-pass Gram-Schmidt in the bm1s inner product) from the unit vector q0;
        returns (Q, H, seconds per step list)."""
        o = self.o
        w = o.bm1s()
        nrm = np.sqrt(o.inner(q0, q0, w))
        Q = [tuple(a / nrm for a in q0)]
        H = np.zeros((k + 1, k)) if H is None else H
        times = []
        for j in range(k):
            t0 = time.perf_counter()
            f = list(self.matvec(Q[j]))
            for _ in range(2):
                for i in range(j + 1):
                    a = o.inner(f, Q[i], w)
                    f = [fc - a * qc for fc, qc in zip(f, Q[i])]
                    H[i, j] += a
            beta = np.sqrt(o.inner(f, f, w))
            H[j + 1, j] = beta
            Q.append(tuple(a / beta for a in f))
            times.append(time.perf_counter() - t0)
        return Q, H, times
