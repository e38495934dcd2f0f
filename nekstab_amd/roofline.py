"""Algorithmic-byte accounting of one matvec: every distinct input array once and every output array
once per kernel invocation, fp64 = 8 B, fp32 = 4 B, indices 4 B, irrespective of cache residency, summed over the
invocations actually executed (iteration counts are run-time data, read from ``nsk_get_stats``).

    bytes_per_matvec = nsteps * [K1 + K2 + K4 + n_helm * d * K3 + n_pres * (precond + E + GMRES) + K10 + projection]
                       + Krylov projection at Arnoldi step j

``d`` = number of velocity components.  The per-kernel entries below are what DESIGN.md section 5 tabulates; they are
the *compulsory* traffic of the launch-per-kernel formulation, so a kernel that keeps state in registers across iterations
(the persistent velocity solve) moves fewer bytes than its algorithmic figure -- which is the point of it.
"""
from __future__ import annotations


def _zero_counts(zero_arrays, ndim):
    """(metric terms, G factors, base-flow constants) that are zero on every node and not loaded (nsk_stats.zero_arrays; hexahedra)."""
    if ndim != 3 or not zero_arrays:
        return 0, 0, 0
    z = int(zero_arrays)
    return bin(z & 0x1ff).count("1"), bin((z >> 9) & 0x7).count("1"), bin((z >> 12) & 0xfff).count("1")


def per_step_bytes(*, nel, lx1, ndim=2, nvert=0, coarse_lda=0, patch_stride=0, nproj=0, helm_iters=0.0, pres_iters=0.0,
                   pres_jsum=None, coarse_bytes=None, gs_lag=1, zero_arrays=0, fuse2=0, tc_cols=20):
    """Algorithmic bytes of ONE time step, by kernel family.  ``helm_iters`` / ``pres_iters``: mean iterations per step
    (all velocity components advance together in one CG iteration).  Hexahedra (``ndim = 3``): ``pres_jsum`` = mean per step of
    the sum over the GMRES columns of their basis index j (nsk_stats.total_pres_jsum / total_steps; the Gram-Schmidt bytes
    are proportional to it), ``coarse_bytes`` = nsk_stats.coarse_bytes_per_solve, ``gs_lag``: the lagged Gram-Schmidt
    sequence (two basis reads per column) or the classic one (four).  ``zero_arrays``: nsk_stats.zero_arrays -- arrays that vanish on
    every node are not inputs of the kernels (Nek5000 skips them on its undeformed elements the same way) and are not counted."""
    zm, zg, zb = _zero_counts(zero_arrays, ndim)
    d = ndim
    N, M, ND = lx1, lx1 - 2, 3 * lx1 // 2
    P, P2, Pd = nel * N ** d, nel * M ** d, nel * ND ** d
    f = 8.0
    nmet2 = d * d - zm                  # Gauss-mesh metrics per pressure point
    out = {}
    # K1 convection + sponge: u' (d), spng, bm1, bf out (d) on P; base-flow constants on the dealiasing mesh: 2-D 6, 3-D 12
    out["K1 convect"] = f * (P * (2 * d + 2) + Pd * (6 if d == 2 else 12 - zb))
    # K2 rhs: u (d), dulag (3d), bf (d), exlag rw (2d + 2d), ulag rw (2d + 2d), bm1, G factors (3 | 6); p, plag rw, pext, metrics on P2; rloc, bloc out (2d)
    ng = 3 if d == 2 else 6 - zg
    out["K2 rhs"] = f * (P * (d + 3 * d + d + 4 * d + 4 * d + 1 + ng + 2 * d) + P2 * (4 + nmet2))
    # K3+K4+K5 one CG iteration of one component: SURVEY 8(d) table: 148 B/pt (2-D), 172 B/pt (3-D)
    out["K3 helm iteration (x n_helm x d)"] = (148.0 if d == 2 else 172.0 - f * zg) * P * d * helm_iters
    # K4' pressure rhs: hx (d), dulag rw (3d + 3d), u rw (2d); metrics, V0 out, PX reads on P2
    out["K4 pres_rhs"] = f * (P * (d + 6 * d + 2 * d) + P2 * (nmet2 + 1 + nproj))
    # per GMRES iteration (mean basis index j ~ (n-1)/2 unless the sum of the basis indices was logged)
    jbar = max(pres_iters - 1.0, 0.0) / 2.0
    jsum = pres_jsum if pres_jsum is not None else jbar * pres_iters
    if d == 2:
        coarse = 4.0 * nvert * coarse_lda + f * (nel * 2 ** d + 2 * nvert)                  # fp32 dense inverse + restriction + x_c
        schwarz = 4.0 * nel * patch_stride * M ** d + 4.0 * nel * patch_stride + f * nel * patch_stride + f * P2 * (1 + nmet2) + f * P * d
        divgs = f * (P * (d + 1) + 2.0 * P) + f * P2 * (nmet2 + 1) + f * P2 * (jsum + pres_iters)   # yl gather (d), binv, 16-B table; metrics, w out; V_0..j for the dots
        gupd = f * P2 * (jsum + 3.0 * (pres_iters + 1.0))                                  # V_0..j, V_{j+1} read + write
        if fuse2:
            # round 6, two launches per iteration.  k_schwarz_uc = coarse workgroups (dense inverse, corner slots) + Schwarz workgroups that
            # form v_j on the fly: raw w and V_0..j-1 at the PS patch nodes of every element, V_j and the Schwarz part of Z_j out;
            # k_divgs_t = E apply + the coarse image Tc (tc_cols fp64 columns per pressure node) + Z_j read / write + V_0..j for the dots.
            # No k_gmres_update per iteration (the column is closed inside k_schwarz_uc; the solve starts inside its first launch).
            jb = jsum / pres_iters if pres_iters > 0 else 0.0
            schwarz = schwarz + f * nel * patch_stride * jb + f * P2
            divgs_it = f * (P * (d + 1) + 2.0 * P) + f * P2 * (nmet2 + 1) + f * P2 * (jb + 1) + f * nel * tc_cols * M ** d + 4.0 * nel * tc_cols + 2.0 * f * P2
            out["K6 coarse (x n_pres)"] = coarse * pres_iters
            out["K6 schwarz (x n_pres)"] = schwarz * pres_iters
            out["K7 divgs (x n_pres)"] = divgs_it * pres_iters
            out["K7 gmres_update (x n_pres)"] = 0.0
        else:
            out["K6 coarse (x n_pres)"] = coarse * pres_iters
            out["K6 schwarz (x n_pres)"] = schwarz * pres_iters
            out["K7 divgs (x n_pres)"] = divgs if pres_jsum is not None else f * (P * (d + 1) + 2.0 * P) * pres_iters + f * P2 * (nmet2 + 1 + (jbar + 1)) * pres_iters
            out["K7 gmres_update (x n_pres)"] = gupd if pres_jsum is not None else f * P2 * (jbar + 3) * (pres_iters + 1.0)
    else:
        # hexahedra.  coarse solve: its operator storage (block-circulant inverses / dense inverse / degree x sparse rows:
        # nsk_stats.coarse_bytes_per_solve) + corner restrictions (8 per element, value + index) + two vertex vectors
        coarse = (coarse_bytes or 0.0) + nel * 8 * (f + 4.0) + 2.0 * f * nvert
        # Schwarz by fast diagonalisation + D^T: V_j (patch gather: the basis vector once), 4-B patch table on P, 3 N^2 + 3 N factors per
        # element, nine metrics, corner prolongation (8 x (value + index) per element); Z_j and yl (d) out
        schwarz = f * P2 + 4.0 * P + f * nel * (3 * N * N + 3 * N) + f * P2 * nmet2 + nel * 8 * (f + 4.0) + f * P2 + f * P * d
        # E apply: yl (d) gathered through the 16-B table, B^-1, nine metrics, w out
        divgs = f * P * d + 16.0 * P + f * P + f * P2 * nmet2 + f * P2
        if gs_lag:
            # k_gs_dots: V_0..j and w; k_gs_lag: V_0..j and w again, w' out, the pending v_j out, 8 corner restrictions per element
            gs = f * P2 * ((jsum + 2.0 * pres_iters) + (jsum + 4.0 * pres_iters)) + f * nel * 8 * pres_iters
        else:
            # dots inside k_divgs (j + 1), k_gmres_reorth (w, V twice, w' out), k_gmres_update (w', V, v out) + corner restrictions
            gs = f * P2 * (4.0 * jsum + 8.0 * pres_iters) + f * nel * 8 * pres_iters
        out["K6 coarse (x n_pres)"] = coarse * pres_iters
        out["K6 schwarz (x n_pres)"] = schwarz * pres_iters
        out["K7 divgs (x n_pres)"] = divgs * pres_iters
        out["K7 gram-schmidt (x n_pres)"] = gs + f * P2 * 2.0                              # + normalising V_0 once per solve
    # K10 pressure / velocity update + projection space
    out["K10 pres_update"] = f * (P2 * (pres_iters + nproj + 3 + nmet2) + P * d)
    out["K10 vel_update(+proj)"] = f * (P * (d + 1 + 2 * d + 2.0) + P2 * (nmet2 + 2 + 2 * nproj))
    out["projection apply/update"] = f * P2 * (2 + nproj + 4 * nproj + 2) if nproj else 0.0
    return out


def helm_launch_bytes(*, nel, lx1, ndim, zero_arrays=0):
    """One Helmholtz CG launch (ALL components): (SURVEY 8(d) per-component rule, distinct arrays counted once).  The rule
    counts the arrays the components share -- geometric factors, mass, mask, multiplicity, the 16-B gather table -- once per
    component; the launch reads them once."""
    P = nel * lx1 ** ndim
    zg = _zero_counts(zero_arrays, ndim)[1]
    rule = (148.0 if ndim == 2 else 172.0 - 8.0 * zg) * P * ndim
    ng = 3 if ndim == 2 else 6 - zg
    per_comp = 8.0 * (4 * 2 + 2 + 1)                    # x, r, p, s read + write; A z of the last iteration in, of this one out; Jacobi diagonal
    shared = 8.0 * (ng + 3) + 16.0                      # G factors, mass, mask, 1 / multiplicity; gather table
    return rule, P * (per_comp * ndim + shared)


def krylov_bytes(nstate, j):
    """Two-pass classical Gram-Schmidt at Arnoldi step j: read Q(1:j) for the dots, again for the update,
    f read + write, per pass."""
    return 2.0 * (2.0 * j + 2.0) * nstate * 8.0


def matvec_bytes(stats_total, nsteps, **geom):
    """Bytes of one matvec from the accumulated solver statistics (``total_*`` fields of nsk_get_stats)."""
    steps = max(stats_total["total_steps"], 1)
    if geom.get("ndim", 2) == 3:
        geom = dict(geom, pres_jsum=stats_total.get("total_pres_jsum", 0) / steps, coarse_bytes=stats_total.get("coarse_bytes_per_solve", 0.0))
    per = per_step_bytes(helm_iters=stats_total["total_helm_iters"] / steps, pres_iters=stats_total["total_pres_iters"] / steps, **geom)
    return nsteps * sum(per.values()), per
