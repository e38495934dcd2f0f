"""Element-sharded execution: partition of the elements over ranks and a backend
with the NekStabHip vector interface whose vectors are lists of per-rank device vectors.

``ShardGroup`` drives R *virtual ranks* inside one process (lock-step launches, loop-back copies
as transport) -- the mode the single-GPU tests use to prove that the sharded path reproduces the
single-rank result.  With one process per GPU the same shard contexts talk over RCCL
(``nsk_comm_init_rccl``; not exercised on the 1-GPU test box).
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from .capi import NekStabHip, NskError, _dp


def partition_rcb(case, nranks: int) -> np.ndarray:
    """Recursive coordinate bisection of element centroids -> owner rank per element.
    (The reference partitions with the .ma2 RSB tree; any partition of whole elements works.)"""
    ax = tuple(range(1, case.x.ndim))
    cen = [case.x.mean(axis=ax), case.y.mean(axis=ax)] + ([case.z.mean(axis=ax)] if getattr(case, "ndim", 2) == 3 else [])
    part = np.zeros(case.nel, dtype=np.int32)

    def rec(idx, r0, nr):
        if nr == 1:
            part[idx] = r0
            return
        nl = nr // 2
        key = max(cen, key=lambda a: np.ptp(a[idx]))[idx]          # split along the longest extent
        order = idx[np.argsort(key, kind="stable")]
        cut = len(order) * nl // nr
        rec(order[:cut], r0, nl)
        rec(order[cut:], r0 + nl, nr - nl)

    rec(np.arange(case.nel), 0, nranks)
    return part


def rank_submesh(case, part, rank: int, rings: int = 2) -> np.ndarray:
    """Global ids (ascending) of the elements ``rank`` owns plus ``rings`` rings of node-sharing neighbours: the sub-mesh a
    rank-local set-up works on (two rings: the assembled mass of every node of every element that shares a node with an
    owned element is complete, which is what the Schwarz patches and the coarse rows of the owned elements reach)."""
    gid = np.asarray(case.gid).reshape(case.nel, -1)
    sel = np.asarray(part) == rank
    for _ in range(rings):
        nodes = np.zeros(int(case.nglob), dtype=bool)
        nodes[gid[sel].ravel()] = True
        sel = nodes[gid].any(axis=1)
    return np.where(sel)[0]


def velocity_halo_plan(gid_rows, owner_rows, rank: int) -> dict:
    """The dssum interface of ``rank`` as the library derives it (nsk_shard.inc: shard_fill): {peer: ascending global node ids
    that elements of ``rank`` AND elements of ``peer`` touch}.  ``gid_rows`` [n, nodes] / ``owner_rows`` [n]: the elements this
    rank knows -- the whole mesh, or its rank-local sub-mesh (own elements + rings), which must give the same plan.  One message
    per peer carries one partial sum per listed node (per component), in this order on both sides."""
    gid_rows = np.asarray(gid_rows).reshape(len(owner_rows), -1)
    owner_rows = np.asarray(owner_rows)
    mine = np.unique(gid_rows[owner_rows == rank])
    plan = {}
    for p in np.unique(owner_rows):
        if p == rank:
            continue
        shared = np.intersect1d(mine, np.unique(gid_rows[owner_rows == p]), assume_unique=True)
        if len(shared):
            plan[int(p)] = shared
    return plan


def subset_case(case, sub):
    """The case restricted to the elements ``sub`` (global node / vertex ids and nglob / nvert kept)."""
    import dataclasses
    kw = {}
    for f in dataclasses.fields(case):
        v = getattr(case, f.name)
        if not isinstance(v, np.ndarray):
            continue
        if f.name == "ub":
            kw[f.name] = v[:, sub]
        elif v.shape[0] == case.nel:
            kw[f.name] = v[sub]
    meta = dict(case.meta)
    meta["vert"] = np.asarray(case.meta["vert"]).reshape(case.nel, -1)[sub]
    return dataclasses.replace(case, nel=len(sub), meta=meta, **kw)


class LocalParent(NekStabHip):
    """Rank-local set-up (nsk_init_local ... nsk_local_finish): the context of ONE rank's sub-mesh.  What Nek5000 does by
    construction -- every MPI rank sets up its own elements -- instead of the whole-mesh parent every rank of ``ShardRank``
    used to build.  ``finish`` takes the four scalars and the coarse rows reduced / gathered over all ranks."""

    def __init__(self, case, part, rank: int, **kw):
        part = np.asarray(part)
        self.rank = rank
        self.sub = np.ascontiguousarray(rank_submesh(case, part, rank), dtype=np.int64)
        self.part_sub = np.ascontiguousarray(part[self.sub], dtype=np.int32)
        sc = subset_case(case, self.sub)
        super().__init__(sc, sc.meta["vert"], sc.meta["nvert"], local_own=(self.part_sub == rank), **kw)
        vol, ct, lm = C.c_double(), C.c_double(), C.c_double()
        npo, nr = C.c_longlong(), C.c_longlong()
        self._chk(self.lib.nsk_local_info(self.ctx, C.byref(vol), C.byref(ct), C.byref(lm), C.byref(npo), C.byref(nr)))
        self.vol_own, self.ctarg, self.fd_lmax, self.npr_own = vol.value, ct.value, lm.value, npo.value
        n = nr.value
        self.rows_u, self.rows_v, self.rows_a = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        ip = C.POINTER(C.c_int)
        self._chk(self.lib.nsk_local_rows(self.ctx, self.rows_u.ctypes.data_as(ip), self.rows_v.ctypes.data_as(ip), self.rows_a.ctypes.data_as(_dp)))

    def finish(self, vol, ctarg, fd_lmax, npr_glob, u, v, a):
        u, v = np.ascontiguousarray(u, dtype=np.int32), np.ascontiguousarray(v, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        ip = C.POINTER(C.c_int)
        self._chk(self.lib.nsk_local_finish(self.ctx, float(vol), float(ctarg), float(fd_lmax), int(npr_glob), len(a),
                                            u.ctypes.data_as(ip), v.ctypes.data_as(ip), a.ctypes.data_as(_dp)))
        dt, ns = C.c_double(), C.c_int()
        x, y, z = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._chk(self.lib.nsk_get_info(self.ctx, C.byref(dt), C.byref(ns), C.byref(x), C.byref(y), C.byref(z)))
        self.dt, self.nsteps = dt.value, ns.value
        self.rows_u = self.rows_v = self.rows_a = None

    def finish_dist(self, dist):
        """``finish`` with the exchange done over torch.distributed (any backend)."""
        self.finish(*exchange_local(dist, self.vol_own, self.npr_own, self.ctarg, self.fd_lmax, self.rows_u, self.rows_v, self.rows_a))


def exchange_local(dist, vol_own, npr_own, ctarg, fd_lmax, u, v, a):
    """What the ranks of a rank-local set-up tell each other, over torch.distributed (host tensors for gloo, device tensors
    for nccl): sums of the owned volume and pressure dofs, maxima of the CFL number and of the fast-diagonalisation
    eigenvalue, and the coarse rows of all ranks concatenated in rank order (every rank fills its slice of a zeroed array
    and the arrays are summed: one collective per array on any backend).  Returns the arguments of ``LocalParent.finish``."""
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    rank, world = dist.get_rank(), dist.get_world_size()
    s = torch.tensor([vol_own, float(npr_own)], dtype=torch.float64, device=dev)
    m = torch.tensor([ctarg, fd_lmax], dtype=torch.float64, device=dev)
    dist.all_reduce(s)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    cnt = torch.zeros(world, dtype=torch.int64, device=dev)
    cnt[rank] = len(a)
    dist.all_reduce(cnt)
    off = np.concatenate([[0], np.cumsum(cnt.cpu().numpy())])
    lo, hi, tot = int(off[rank]), int(off[rank + 1]), int(off[-1])
    out = []
    for src, dt in ((u, torch.int32), (v, torch.int32), (a, torch.float64)):
        T = torch.zeros(tot, dtype=dt, device=dev)
        T[lo:hi] = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
        dist.all_reduce(T)
        out.append(T.cpu().numpy())
    s, m = s.cpu().numpy(), m.cpu().numpy()
    return float(s[0]), float(m[0]), float(m[1]), int(round(s[1])), out[0], out[1], out[2]


def local_parents(case, nranks: int, part=None, **kw):
    """The rank-local set-up of all ``nranks`` ranks inside ONE process (virtual ranks: what the single-GPU tests use), with
    the exchange done by plain loops.  Returns (parents, part)."""
    part = np.ascontiguousarray(partition_rcb(case, nranks) if part is None else part, dtype=np.int32)
    P = [LocalParent(case, part, r, **kw) for r in range(nranks)]
    vol = sum(p.vol_own for p in P)
    npr = sum(p.npr_own for p in P)
    ct = max(p.ctarg for p in P)
    lm = max(p.fd_lmax for p in P)
    u, v, a = (np.concatenate([getattr(p, k) for p in P]) for k in ("rows_u", "rows_v", "rows_a"))
    for p in P:
        p.finish(vol, ct, lm, npr, u, v, a)
    return P, part


def _shard_elems(lib, ctx, n):
    out = np.empty(n, dtype=np.int64)
    rc = lib.nsk_shard_elems(ctx, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    if rc != 0:
        raise NskError(rc, lib.nsk_last_error().decode())
    return out


def shard_halo_counts(lib, ctx, nranks):
    """(vel, pres_send, pres_recv) per peer rank of one shard context: nsk_shard_halo_counts."""
    arrs = [np.zeros(nranks, dtype=np.int32) for _ in range(3)]
    rc = lib.nsk_shard_halo_counts(ctx, *[a.ctypes.data_as(C.POINTER(C.c_int)) for a in arrs])
    if rc != 0:
        raise NskError(rc, lib.nsk_last_error().decode())
    return arrs


class ShardVec:
    def __init__(self, parts):
        self.parts = parts            # one device handle per rank


class ShardGroup:
    """R virtual ranks of one full-mesh context; same interface as NekStabHip for krylov.py."""

    def __init__(self, full, case, nranks: int, part=None):
        """``full``: the whole-mesh context the shards are cut from, or the list of the ranks' ``LocalParent`` contexts
        (rank-local set-up, ``local_parents``)."""
        self.parents = list(full) if isinstance(full, (list, tuple)) else None
        if self.parents is not None:
            full = self.parents[0]
        self.full, self.lib, self.R = full, full.lib, nranks
        self.part = np.ascontiguousarray(partition_rcb(case, nranks) if part is None else part, dtype=np.int32)
        self.nsteps, self.dt = full.nsteps, full.dt
        ndim = int(getattr(case, "ndim", 2))
        self.nel, self.lx1, self.lx2 = case.nel, full.lx1, full.lx2
        self.npres, self.nvel = case.nel * self.lx2 ** ndim, case.nel * self.lx1 ** ndim
        self.elems = [np.where(self.part == r)[0] for r in range(nranks)]
        self.ctx = []
        for r in range(nranks):
            out = C.c_void_p()
            if self.parents is None:
                self._chk(self.lib.nsk_shard_create(full.ctx, self.part.ctypes.data_as(C.POINTER(C.c_int)), r, nranks, C.byref(out)))
            else:
                p = self.parents[r]
                self._chk(self.lib.nsk_shard_create_local(p.ctx, p.part_sub.ctypes.data_as(C.POINTER(C.c_int)),
                                                          p.sub.ctypes.data_as(C.POINTER(C.c_longlong)), r, nranks, C.byref(out)))
                if r:
                    self._chk(self.lib.nsk_shard_share_stream(out, self.ctx[0]))     # virtual ranks run in stream order
            self.ctx.append(out)
            self.elems[r] = _shard_elems(self.lib, out, len(self.elems[r]))      # the shard's own order: boundary elements first
        self._arr = (C.c_void_p * nranks)(*[c.value for c in self.ctx])

    def _chk(self, rc):
        if rc != 0:
            raise NskError(rc, self.lib.nsk_last_error().decode())

    def close(self):
        for c in self.ctx:
            self.lib.nsk_finalize(c)
        self.ctx = []

    def release_parent(self):
        """Free the parent's element-major device arrays (nsk_shard_release_parent): the GPU then holds the shards only."""
        for p in (self.parents or [self.full]):
            self._chk(self.lib.nsk_shard_release_parent(p.ctx))

    # ---- vectors
    def alloc(self, n=1):
        per = []
        for c in self.ctx:
            arr = (C.c_void_p * n)()
            self._chk(self.lib.nsk_vec_alloc(c, n, arr))
            per.append([C.c_void_p(arr[i]) for i in range(n)])
        return [ShardVec([per[r][i] for r in range(self.R)]) for i in range(n)]

    def free(self, vecs):
        for r, c in enumerate(self.ctx):
            arr = (C.c_void_p * len(vecs))(*[v.parts[r].value for v in vecs])
            self._chk(self.lib.nsk_vec_free(c, len(vecs), arr))

    def upload3(self, v, vx, vy, vz, pr):
        for r, c in enumerate(self.ctx):
            e = self.elems[r]
            a, b, w, p = (np.ascontiguousarray(np.asarray(f).reshape(self.nel, -1)[e], dtype=np.float64) for f in (vx, vy, vz, pr))
            self._chk(self.lib.nsk_vec_upload3(c, v.parts[r], a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), w.ctypes.data_as(_dp), p.ctypes.data_as(_dp)))

    def download3(self, v):
        n, m = self.lx1, self.lx2
        out = [np.empty((self.nel, n, n, n)) for _ in range(3)] + [np.empty((self.nel, m, m, m))]
        for r, c in enumerate(self.ctx):
            e = self.elems[r]
            loc = [np.empty((len(e), n, n, n)) for _ in range(3)] + [np.empty((len(e), m, m, m))]
            self._chk(self.lib.nsk_vec_download3(c, v.parts[r], *[a.ctypes.data_as(_dp) for a in loc]))
            for o, a in zip(out, loc):
                o[e] = a
        return tuple(out)

    def upload(self, v, vx, vy, pr):
        for r, c in enumerate(self.ctx):
            e = self.elems[r]
            a, b, p = (np.ascontiguousarray(np.asarray(f).reshape(self.nel, -1)[e], dtype=np.float64) for f in (vx, vy, pr))   # flat or (nel, ...) arrays
            self._chk(self.lib.nsk_vec_upload(c, v.parts[r], a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), p.ctypes.data_as(_dp)))

    def download(self, v):
        n, m = self.lx1, self.lx2
        vx = np.empty((self.nel, n, n)); vy = np.empty((self.nel, n, n)); pr = np.empty((self.nel, m, m))
        for r, c in enumerate(self.ctx):
            e = self.elems[r]
            a = np.empty((len(e), n, n)); b = np.empty((len(e), n, n)); p = np.empty((len(e), m, m))
            self._chk(self.lib.nsk_vec_download(c, v.parts[r], a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), p.ctypes.data_as(_dp)))
            vx[e], vy[e], pr[e] = a, b, p
        return vx, vy, pr

    def group_test(self, which, field):
        """kernel-level check across shards: which=0 dssum (velocity mesh), 1 E-apply (pressure mesh)"""
        ins = [np.ascontiguousarray(field[e], dtype=np.float64) for e in self.elems]
        outs = [np.empty_like(a) for a in ins]
        ia = (_dp * self.R)(*[a.ctypes.data_as(_dp) for a in ins])
        oa = (_dp * self.R)(*[a.ctypes.data_as(_dp) for a in outs])
        self._chk(self.lib.nsk_group_test(self._arr, self.R, which, ia, oa))
        full = np.empty_like(np.asarray(field, dtype=np.float64))
        for e, o in zip(self.elems, outs):
            full[e] = o
        return full

    # ---- operator
    def matvec(self, f, q, mode=0):
        fa = (C.c_void_p * self.R)(*[p.value for p in f.parts])
        qa = (C.c_void_p * self.R)(*[p.value for p in q.parts])
        rc = self.lib.nsk_group_matvec(self._arr, self.R, mode, fa, qa)
        if rc != 0:
            from .capi import NskStats
            st = NskStats(); self.lib.nsk_get_stats(self.ctx[0], C.byref(st))
            raise NskError(rc, self.lib.nsk_last_error().decode() + " stats=" + str({f: getattr(st, f) for f, _ in NskStats._fields_}))

    def nonlinear_map(self, f, q, subtract_q=False):
        """Phi_T(q) of the full equations on the shards (nonlinear_forward_map, core/newton_krylov.f:336-378)."""
        fa = (C.c_void_p * self.R)(*[p.value for p in f.parts])
        qa = (C.c_void_p * self.R)(*[p.value for p in q.parts])
        self._chk(self.lib.nsk_group_nonlinear_map(self._arr, self.R, fa, qa, int(bool(subtract_q))))

    def _refresh_info(self):
        dt, ns = C.c_double(), C.c_int()
        self._chk(self.lib.nsk_get_info(self.ctx[0], C.byref(dt), C.byref(ns), None, None, None))
        self.dt, self.nsteps = dt.value, ns.value

    def set_baseflow(self, q):
        """New linearisation point on every shard; dt / nsteps from the CFL maximum over all ranks."""
        qa = (C.c_void_p * self.R)(*[p.value for p in q.parts])
        self._chk(self.lib.nsk_group_set_baseflow(self._arr, self.R, qa))
        self._refresh_info()

    def set_orbit(self, q0, spng_str=0.0, end=None):
        """Time-periodic base flow stored per shard (Floquet, core/matvec.f:191-236)."""
        qa = (C.c_void_p * self.R)(*[p.value for p in q0.parts])
        ea = (C.c_void_p * self.R)(*[p.value for p in end.parts]) if end is not None else None
        self._chk(self.lib.nsk_group_set_orbit(self._arr, self.R, qa, float(spng_str), ea))
        self._refresh_info()

    def stats(self):
        from .capi import NskStats
        st = NskStats()
        self._chk(self.lib.nsk_get_stats(self.ctx[0], C.byref(st)))
        return {f: getattr(st, f) for f, _ in NskStats._fields_}

    def set_nsteps(self, n):
        for c in self.ctx:
            self._chk(self.lib.nsk_set_nsteps(c, n))
        self.nsteps = n

    def set_option(self, name, value):
        for c in self.ctx:
            self._chk(self.lib.nsk_set_option(c, name.encode(), float(value)))

    # ---- vector algebra: rank-local kernels + a sum over ranks (all-reduce when ranks are processes)
    def _dots(self, f, Q):
        tot = np.zeros(len(Q))
        for r, c in enumerate(self.ctx):
            arr = (C.c_void_p * len(Q))(*[v.parts[r].value for v in Q])
            out = np.zeros(len(Q))
            self._chk(self.lib.nsk_local_dots(c, f.parts[r], arr, len(Q), out.ctypes.data_as(_dp)))
            tot += out
        return tot

    def dot(self, p, q):
        return float(self._dots(p, [q])[0])

    def norm(self, p):
        return float(np.sqrt(self.dot(p, p)))

    def scal(self, p, a):
        for r, c in enumerate(self.ctx):
            self._chk(self.lib.nsk_scal(c, p.parts[r], a))

    def axpy(self, p, a, q):
        for r, c in enumerate(self.ctx):
            self._chk(self.lib.nsk_axpy(c, p.parts[r], a, q.parts[r]))

    def copy(self, dst, src):
        for r, c in enumerate(self.ctx):
            self._chk(self.lib.nsk_copy(c, dst.parts[r], src.parts[r]))

    def zero(self, p):
        for r, c in enumerate(self.ctx):
            self._chk(self.lib.nsk_zero(c, p.parts[r]))

    def orth(self, f, Q):
        """update_hessenberg_matrix (core/krylov_decomposition.f:116-202): two projection passes
        with globally summed coefficients, then normalisation."""
        h = np.zeros(len(Q))
        for _ in range(2):
            if not Q:
                break
            cpass = self._dots(f, Q)
            for r, c in enumerate(self.ctx):
                arr = (C.c_void_p * len(Q))(*[v.parts[r].value for v in Q])
                self._chk(self.lib.nsk_project_out(c, f.parts[r], arr, len(Q), cpass.ctypes.data_as(_dp)))
            h += cpass
        beta = self.norm(f)
        self.scal(f, 1.0 / beta)
        return h, beta

    def basis_gemm(self, Q, Z):
        k = len(Q)
        Zc = np.asfortranarray(Z, dtype=np.float64)
        for r, c in enumerate(self.ctx):
            arr = (C.c_void_p * k)(*[v.parts[r].value for v in Q])
            self._chk(self.lib.nsk_basis_gemm(c, arr, k, Zc.ctypes.data_as(_dp), Zc.shape[0]))

    def basis_gemv(self, Q, y, re, im=None):
        k = len(Q)
        yr = np.ascontiguousarray(np.real(y), dtype=np.float64)
        yi = np.ascontiguousarray(np.imag(y), dtype=np.float64)
        for r, c in enumerate(self.ctx):
            arr = (C.c_void_p * k)(*[v.parts[r].value for v in Q])
            self._chk(self.lib.nsk_basis_gemv(c, arr, k, yr.ctypes.data_as(_dp), yi.ctypes.data_as(_dp) if im is not None else None,
                                              re.parts[r], im.parts[r] if im is not None else None))


class ShardRank:
    """One rank of an element-sharded run, one process per GPU; halos and reductions over RCCL.

    Every process builds the full-mesh parent context on its own GPU (replicated set-up), cuts its
    shard (``release_parent()`` then frees the parent's element-major arrays), and joins the communicator whose id rank 0 created (``unique_id`` is exchanged by the
    caller, e.g. with ``torch.distributed.broadcast``).  Same vector interface as NekStabHip, with
    vectors holding this rank's elements only."""

    def __init__(self, full: NekStabHip, case, rank: int, nranks: int, unique_id: bytes | None, part=None):
        self.full, self.lib, self.rank, self.nranks = full, full.lib, rank, nranks
        self.part = np.ascontiguousarray(partition_rcb(case, nranks) if part is None else part, dtype=np.int32)
        self.elems = np.where(self.part == rank)[0]
        self.nsteps, self.dt = full.nsteps, full.dt
        self.lx1, self.lx2 = full.lx1, full.lx2
        self.nel = len(self.elems)
        self.ndim = int(getattr(case, "ndim", 2))
        self.npres, self.nvel = self.nel * self.lx2 ** self.ndim, self.nel * self.lx1 ** self.ndim
        self.ctx = C.c_void_p()
        if isinstance(full, LocalParent):                          # rank-local set-up: `full` covers this rank's sub-mesh only
            self._chk(self.lib.nsk_shard_create_local(full.ctx, full.part_sub.ctypes.data_as(C.POINTER(C.c_int)),
                                                      full.sub.ctypes.data_as(C.POINTER(C.c_longlong)), rank, nranks, C.byref(self.ctx)))
        else:
            self._chk(self.lib.nsk_shard_create(full.ctx, self.part.ctypes.data_as(C.POINTER(C.c_int)), rank, nranks, C.byref(self.ctx)))
        self.elems = _shard_elems(self.lib, self.ctx, self.nel)        # the shard's own order: boundary elements first
        if unique_id is not None:
            buf = C.create_string_buffer(bytes(unique_id), 128)
            self._chk(self.lib.nsk_comm_init_rccl(self.ctx, C.cast(buf, C.c_void_p)))
        self._one = (C.c_void_p * 1)(self.ctx.value)

    @staticmethod
    def new_unique_id(lib) -> bytes:
        buf = C.create_string_buffer(128)
        rc = lib.nsk_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise NskError(rc, lib.nsk_last_error().decode())
        return buf.raw

    def _chk(self, rc):
        if rc != 0:
            raise NskError(rc, self.lib.nsk_last_error().decode())

    def close(self):
        if self.ctx:
            self.lib.nsk_finalize(self.ctx)
            self.ctx = C.c_void_p()

    def release_parent(self):
        """Free the full-mesh parent's element-major device arrays (nsk_shard_release_parent): from here on this rank's GPU
        holds its shard, the 1-D bases and the replicated coarse operator only."""
        self._chk(self.lib.nsk_shard_release_parent(self.full.ctx))

    def alloc(self, n=1):
        arr = (C.c_void_p * n)()
        self._chk(self.lib.nsk_vec_alloc(self.ctx, n, arr))
        return [C.c_void_p(arr[i]) for i in range(n)]

    def free(self, vecs):
        arr = (C.c_void_p * len(vecs))(*[v.value for v in vecs])
        self._chk(self.lib.nsk_vec_free(self.ctx, len(vecs), arr))

    def upload(self, v, vx, vy, pr):
        """vx, vy, pr: full-mesh arrays; this rank keeps its own elements."""
        a, b, p = (np.ascontiguousarray(np.asarray(f).reshape(len(self.part), -1)[self.elems], dtype=np.float64) for f in (vx, vy, pr))
        self._chk(self.lib.nsk_vec_upload(self.ctx, v, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), p.ctypes.data_as(_dp)))

    def upload3(self, v, vx, vy, vz, pr):
        """hexahedral contexts: full-mesh arrays, this rank keeps its own elements."""
        a, b, w, p = (np.ascontiguousarray(np.asarray(f).reshape(len(self.part), -1)[self.elems], dtype=np.float64) for f in (vx, vy, vz, pr))
        self._chk(self.lib.nsk_vec_upload3(self.ctx, v, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), w.ctypes.data_as(_dp), p.ctypes.data_as(_dp)))

    def download3_local(self, v):
        n, m = self.lx1, self.lx2
        loc = [np.empty((self.nel, n, n, n)) for _ in range(3)] + [np.empty((self.nel, m, m, m))]
        self._chk(self.lib.nsk_vec_download3(self.ctx, v, *[a.ctypes.data_as(_dp) for a in loc]))
        return tuple(loc)

    def download_local(self, v):
        n, m = self.lx1, self.lx2
        a = np.empty((self.nel, n, n)); b = np.empty((self.nel, n, n)); p = np.empty((self.nel, m, m))
        self._chk(self.lib.nsk_vec_download(self.ctx, v, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), p.ctypes.data_as(_dp)))
        return a, b, p

    def matvec(self, f, q, mode=0):
        fa = (C.c_void_p * 1)(f.value)
        qa = (C.c_void_p * 1)(q.value)
        self._chk(self.lib.nsk_group_matvec(self._one, 1, mode, fa, qa))

    def nonlinear_map(self, f, q, subtract_q=False):
        fa = (C.c_void_p * 1)(f.value)
        qa = (C.c_void_p * 1)(q.value)
        self._chk(self.lib.nsk_group_nonlinear_map(self._one, 1, fa, qa, int(bool(subtract_q))))

    def _refresh_info(self):
        dt, ns = C.c_double(), C.c_int()
        self._chk(self.lib.nsk_get_info(self.ctx, C.byref(dt), C.byref(ns), None, None, None))
        self.dt, self.nsteps = dt.value, ns.value

    def set_baseflow(self, q):
        qa = (C.c_void_p * 1)(q.value)
        self._chk(self.lib.nsk_group_set_baseflow(self._one, 1, qa))
        self._refresh_info()

    def set_orbit(self, q0, spng_str=0.0, end=None):
        qa = (C.c_void_p * 1)(q0.value)
        ea = (C.c_void_p * 1)(end.value) if end is not None else None
        self._chk(self.lib.nsk_group_set_orbit(self._one, 1, qa, float(spng_str), ea))
        self._refresh_info()

    def stats(self):
        from .capi import NskStats
        st = NskStats()
        self._chk(self.lib.nsk_get_stats(self.ctx, C.byref(st)))
        return {f: getattr(st, f) for f, _ in NskStats._fields_}

    def set_nsteps(self, n):
        self._chk(self.lib.nsk_set_nsteps(self.ctx, n))
        self.nsteps = n

    def set_option(self, name, value):
        self._chk(self.lib.nsk_set_option(self.ctx, name.encode(), float(value)))

    def _dots(self, f, Q):
        arr = (C.c_void_p * len(Q))(*[v.value for v in Q])
        out = np.zeros(len(Q))
        self._chk(self.lib.nsk_local_dots(self.ctx, f, arr, len(Q), out.ctypes.data_as(_dp)))
        self._chk(self.lib.nsk_allreduce_host(self.ctx, out.ctypes.data_as(_dp), len(Q)))     # sum over ranks
        return out

    def dot(self, p, q):
        return float(self._dots(p, [q])[0])

    def norm(self, p):
        return float(np.sqrt(self.dot(p, p)))

    def scal(self, p, a):
        self._chk(self.lib.nsk_scal(self.ctx, p, a))

    def axpy(self, p, a, q):
        self._chk(self.lib.nsk_axpy(self.ctx, p, a, q))

    def copy(self, dst, src):
        self._chk(self.lib.nsk_copy(self.ctx, dst, src))

    def zero(self, p):
        self._chk(self.lib.nsk_zero(self.ctx, p))

    def orth(self, f, Q):
        """update_hessenberg_matrix on sharded vectors: one library call; the two projection passes and the normalisation
        run on the device, the coefficient vectors are all-reduced over the ranks on the stream (RCCL or host-staged)."""
        j = len(Q)
        arr = (C.c_void_p * max(j, 1))(*[v.value for v in Q])
        h = np.zeros(max(j, 1))
        beta = C.c_double()
        self._chk(self.lib.nsk_orth(self.ctx, f, arr, j, h.ctypes.data_as(_dp), C.byref(beta)))
        return h[:j].copy(), beta.value

    def basis_gemm(self, Q, Z):
        k = len(Q)
        Zc = np.asfortranarray(Z, dtype=np.float64)
        arr = (C.c_void_p * k)(*[v.value for v in Q])
        self._chk(self.lib.nsk_basis_gemm(self.ctx, arr, k, Zc.ctypes.data_as(_dp), Zc.shape[0]))


class HostTransport:
    """Host-staged transport for ``nsk_comm_init_host`` over ``torch.distributed`` point-to-point messages and all-reduce
    (any backend that moves CPU tensors, e.g. gloo): what lets several ranks share ONE GPU, so that the exchange protocol
    of the sharded time stepper (pack / unpack tables, message order, reductions) is proven across processes without a
    second GPU.  The C library packs on the device, stages through pinned host memory and calls these two functions."""

    XF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_dp), C.POINTER(_dp))
    AF = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int)

    def __init__(self, dist):
        self.dist = dist
        self.n_exchange = self.n_allreduce = 0
        self.xf = HostTransport.XF(self._exchange)
        self.af = HostTransport.AF(self._allreduce)

    def exchange(self, peers, send):
        """send[k] (1-D float64 array) goes to rank peers[k]; returns what each peer sent back (same lengths)."""
        import torch
        reqs, recv = [], []
        for p, sbuf in zip(peers, send):
            t = torch.from_numpy(np.ascontiguousarray(sbuf, dtype=np.float64).copy())
            r = torch.empty(t.numel(), dtype=torch.float64)
            reqs.append(self.dist.isend(t, int(p)))
            reqs.append(self.dist.irecv(r, int(p)))
            recv.append(r)
        for q in reqs:
            q.wait()
        self.n_exchange += 1
        return [r.numpy() for r in recv]

    def allreduce(self, buf):
        import torch
        t = torch.from_numpy(buf)
        self.dist.all_reduce(t)
        # test hook (tests/test_multiprocess_gpu.py): ONE rank receives the result of all-reduce number NSK_TEST_ALLRED_ULP one
        # ulp off -- what a collective with a rank-dependent reduction order would deliver; the library must refuse the map (NSK_ECOMM)
        hook = os.environ.get("NSK_TEST_ALLRED_ULP")
        if hook is not None and self.n_allreduce == int(hook) and self.dist.get_rank() == self.dist.get_world_size() - 1:
            buf[0] = np.nextafter(buf[0], np.inf)
        self.n_allreduce += 1
        return buf

    def _exchange(self, user, npeers, peers, counts, send, recv):
        try:
            ps = [peers[k] for k in range(npeers)]
            out = self.exchange(ps, [np.ctypeslib.as_array(send[k], shape=(counts[k],)) for k in range(npeers)])
            for k in range(npeers):
                np.ctypeslib.as_array(recv[k], shape=(counts[k],))[:] = out[k]
            return 0
        except Exception as e:      # an exception must not unwind through the C frames
            print("HostTransport.exchange failed:", repr(e), flush=True)
            return -1

    def _allreduce(self, user, buf, n):
        try:
            self.allreduce(np.ctypeslib.as_array(buf, shape=(n,)))
            return 0
        except Exception as e:
            print("HostTransport.allreduce failed:", repr(e), flush=True)
            return -1


def attach_host_transport(shard: "ShardRank", dist):
    """Route the halos and reductions of a ShardRank through ``dist`` (host-staged) instead of RCCL."""
    tr = HostTransport(dist)
    rc = shard.lib.nsk_comm_init_host(shard.ctx, C.cast(tr.xf, C.c_void_p), C.cast(tr.af, C.c_void_p), None)
    if rc != 0:
        raise NskError(rc, shard.lib.nsk_last_error().decode())
    shard._transport = tr          # keep the callbacks alive
    return tr
