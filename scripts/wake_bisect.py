#!/usr/bin/env python3
"""What moves the direct wake branch?  k_dim = 200 direct Arnoldi at lx1 = 6 with converged inner solves,
once for the restated discretisation and once per MODELLING VARIANT -- sponge amplitude / width / ramp,
pressure-extrapolation order, the order ramp of the first steps, dt, Re, base-flow error, inner-solver tolerance -- and, per
variant, the shift s_i of every row of the reference's Spectre_Hd.dat that the reference converged below 1e-7:

    d_i = mu_ref_i - mu_base_i        what separates this build from the reference's table (1e-5 ... 5e-5 on rows 5-23)
    s_i = mu_variant_i - mu_base_i    what the variant does

    corr  = Re <s, d> / (|s| |d|)     1: the variant moves the rows exactly towards the table
    alpha = Re <s, d> / |s|^2         how many times the tried perturbation it would take
    left  = |d - alpha s| / |d|       what is left of the gap at the best amplitude

A variant EXPLAINS the gap if corr ~ 1 and left << 1 at a plausible alpha.  Output: markdown on stdout,
gpurun_out/wake_bisect.json.  The HIP path is the engine (a converged spectrum takes 15 s there, an hour on the numpy oracle);
its converged spectrum equals the oracle's (tests/golden/cylinder_oracle_spectra.npz, tests/test_spectrum_pin_gpu.py).

    python scripts/wake_bisect.py [--only name,name] [--k 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def stepf(x):
    x = np.asarray(x, dtype=float)
    out = np.ones_like(x)
    lo = x <= 0.001
    mid = (~lo) & (x <= 0.999)
    out[lo] = 0.0
    xm = x[mid]
    out[mid] = 1.0 / (1.0 + np.exp(1.0 / (xm - 1.0) + 1.0 / xm))
    return out


def sponge_variant(x, left=5.0, right=5.0, acc=0.333, ramp_by="plateau", amp_l=1.0, amp_r=1.0):
    """spng_set (core/utils.f:235-323) in x only, with knobs: ramp_by = 'plateau' is the reference (the ramp argument is
    divided by the plateau width, so the ramp ends at mth_stepf(0.4993) = 0.5 and jumps to 1), 'ramp' divides by the ramp
    width (a smooth 0 -> 1 rise)."""
    fun = np.zeros_like(x)
    wl, wr, dl, dr = (1 - acc) * left, (1 - acc) * right, acc * left, acc * right
    xmin, xmax = x.min(), x.max()
    xxmax, xxmin = xmax - wr, xmin + wl
    xxmax_c, xxmin_c = xxmax + dr, xxmin - dl
    nl, nr = (wl, wr) if ramp_by == "plateau" else (dl, dr)
    r = np.zeros_like(x)
    m1 = x <= xxmin_c
    m2 = (~m1) & (x < xxmin)
    m4 = (x > xxmax) & (x < xxmax_c)
    m5 = x >= xxmax_c
    r[m1] = amp_l
    r[m2] = amp_l * stepf((xxmin - x[m2]) / nl)
    r[m4] = amp_r * stepf((x[m4] - xxmax) / nr)
    r[m5] = amp_r
    return np.maximum(fun, r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--lx1", type=int, default=6)
    a = ap.parse_args()
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.capi import NekStabHip
    spectre = np.load(os.path.join(GOLDEN, "cylinder_spectre.npz"))["Hd"]
    modes = np.load(os.path.join(GOLDEN, "cylinder_modes.npz"))
    rows = [(n + 1, complex(r[0], r[1]), r[2]) for n, r in enumerate(spectre) if r[2] < 1e-7 and r[1] >= 0]

    def run(case_kw=None, spng=None, opts=None, tol=(1e-13, 1e-6), ub_add=None, ctx_kw=None, xy_file=False):
        case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), a.lx1, **(case_kw or {}))
        if xy_file:                                   # node coordinates as the reference's base-flow file holds them (tests/golden/make_fixtures.py)
            xy = np.load(os.path.join(GOLDEN, "cylinder_bf_xy.npz"))
            assert a.lx1 == 6 and xy["x"].shape == case.x.shape
            print("  (file coordinates - regenerated geometry: max %.2e)" % max(np.abs(xy["x"] - case.x).max(), np.abs(xy["y"] - case.y).max()), file=sys.stderr)
            case.x, case.y = xy["x"].copy(), xy["y"].copy()
        if spng is not None:
            case.spng = spng(case)
        if ub_add is not None:
            case.ub = case.ub + ub_add(case)
        kw = dict(tol_helm=tol[0], tol_pres=tol[1], tol_relative=1, schwarz_layers=2, max_helm_iter=200, max_pres_iter=48, nproj=0)
        kw.update(ctx_kw or {})
        h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], **kw)
        for k, v in (opts or {}).items():
            h.set_option(k, v)
        # the SAME inner product for every variant (the reference's sponge mask), so that only the operator changes
        qx, qy = seed.add_noise(case)
        v0, v1 = h.alloc(2)
        h.upload(v0, qx, qy, np.zeros(h.npres))
        h.scal(v0, 1.0 / h.norm(v0))
        h.matvec(v1, v0, 0)
        t0 = time.time()
        res = krylov.krylov_schur(h, v1, a.k, mode=0, schur_tgt=0)
        wall = time.time() - t0
        ns = h.nsteps
        h.close()
        out = []
        for n, z, rr in rows:
            j = int(np.argmin(np.abs(res.vals - z)))
            out.append((res.vals[j], res.residual[j]))
        return out, ns, wall

    dre = modes["dRe_u"].astype(np.float64)
    dim = modes["dIm_u"].astype(np.float64)
    V = {
        "sponge amplitude x 1.05 (both)": dict(spng=lambda c: sponge_variant(c.x, amp_l=1.05, amp_r=1.05)),
        "right sponge amplitude x 1.05": dict(spng=lambda c: sponge_variant(c.x, amp_r=1.05)),
        "left sponge amplitude x 1.05": dict(spng=lambda c: sponge_variant(c.x, amp_l=1.05)),
        "sponge strength 1.7 (spng_str of the .par applied)": dict(spng=lambda c: sponge_variant(c.x, amp_l=1.7, amp_r=1.7)),
        "sponge ramp divided by its own width (smooth rise to 1)": dict(spng=lambda c: sponge_variant(c.x, ramp_by="ramp")),
        "right sponge length 5 -> 5.1": dict(spng=lambda c: sponge_variant(c.x, right=5.1)),
        "left sponge length 5 -> 5.1": dict(spng=lambda c: sponge_variant(c.x, left=5.1)),
        "no pressure extrapolation (p* = p^n at every order)": dict(opts={"dbg_pext": 0}),
        "pressure extrapolation from step 2 on": dict(opts={"dbg_pext": 2}),
        "Adams-Bashforth 2 at step 2 ([1.5, -0.5] for [2, -1])": dict(opts={"dbg_ab2": 1}),
        "order ramp capped at 2 (BDF2 / EXT2 throughout)": dict(opts={"dbg_max_order": 2}),
        "CFL target 0.5 -> 0.45 (111 steps per period)": dict(case_kw={"cfl": 0.45}),
        "CFL target 0.5 -> 0.55 (91 steps per period)": dict(case_kw={"cfl": 0.55}),
        "Re 50 -> 50.05": dict(case_kw={"re": 50.05}),
        "base flow + 1e-4 x Re(leading eigenmode)": dict(ub_add=lambda c: 1e-4 * dre),
        "base flow + 1e-4 x Im(leading eigenmode)": dict(ub_add=lambda c: 1e-4 * dim),
        "base flow x (1 + 1e-4)": dict(ub_add=lambda c: 1e-4 * c.ub),
        "Helmholtz tolerance 1e-10 (relative)": dict(tol=(1e-10, 1e-6)),
        "Helmholtz tolerance 1e-10, zero initial guess (Nek's cggo starts from 0)": dict(tol=(1e-10, 1e-6), opts={"helm_guess": 0}),
        "Helmholtz tolerance 1e-9, zero initial guess": dict(tol=(1e-9, 1e-6), opts={"helm_guess": 0}),
        "pressure tolerance 1e-3 (relative), Helmholtz converged": dict(tol=(1e-13, 1e-3)),
        "pressure tolerance 1e-2 with 20 projection vectors (mxprev = 20)": dict(tol=(1e-13, 1e-2), ctx_kw={"nproj": 20}),
        # round 4.  (`lxd` is not a free parameter: the reference's SIZE:13 has lxd = 09 = 3 lx1 / 2, what this build uses.)
        "geometry: node coordinates of the base-flow file's X block instead of the regenerated bilinear + arc geometry": dict(xy_file=True),
        "Nek-style solves: ABSOLUTE tolerances 1e-9 / 1e-7 in Nek's norms (1cyl.par:29,34), zero initial guess, >= 1 GMRES iteration, 20 projection vectors": dict(
            tol=(1e-9, 1e-7), ctx_kw={"tol_relative": 0, "nproj": 20}, opts={"helm_guess": 0, "min_pres_iter": 1}),
        "the same with the velocity tolerance 1e-10 absolute": dict(tol=(1e-10, 1e-7), ctx_kw={"tol_relative": 0, "nproj": 20}, opts={"helm_guess": 0, "min_pres_iter": 1}),
    }
    only = [s for s in a.only.split(",") if s]
    base, ns0, w0 = run()
    print("# Direct wake branch at lx1 = %d: sensitivity of the rows of Spectre_Hd.dat to modelling variants\n" % a.lx1)
    print("baseline: restated discretisation, inner solves 1e-13 / 1e-6, k_dim = %d, nsteps = %d, %.0f s\n" % (a.k, ns0, w0))
    print("| row | reference | baseline | residual (ref / ours) | d = ref - baseline |")
    print("|---|---|---|---|---|")
    d = np.array([z - b[0] for (n, z, rr), b in zip(rows, base)])
    for (n, z, rr), b, di in zip(rows, base, d):
        print("| %d | %.7f%+.7fi | %.9f%+.9fi | %.0e / %.0e | %+.2e %+.2ei (|d| = %.1e) |" % (n, z.real, z.imag, b[0].real, b[0].imag, rr, b[1], di.real, di.imag, abs(di)))
    wake = np.array([n > 4 for n, _, _ in rows])
    print("\nrows > 4 (the wake branch): |d| = %.2e (root sum of squares over %d rows); rows 1-4: %.2e\n" % (np.linalg.norm(d[wake]), wake.sum(), np.linalg.norm(d[~wake])))
    print("| variant | nsteps | max |s| rows 1-4 | max |s| wake rows | corr | alpha | left |")
    print("|---|---|---|---|---|---|---|")
    rec = {"rows": [n for n, _, _ in rows], "reference": [[z.real, z.imag] for _, z, _ in rows], "baseline": [[b[0].real, b[0].imag] for b in base], "variants": {}}
    for name, kw in V.items():
        if only and not any(o in name for o in only):
            continue
        try:
            got, ns, wl = run(**kw)
        except Exception as e:                      # noqa: BLE001
            print("| %s | failed: %s |" % (name, str(e)[:80]))
            continue
        s = np.array([g[0] - b[0] for g, b in zip(got, base)])
        sw, dw = s[wake], d[wake]
        dotp = float(np.real(np.vdot(sw, dw)))
        corr = dotp / (np.linalg.norm(sw) * np.linalg.norm(dw) + 1e-300)
        alpha = dotp / (np.linalg.norm(sw) ** 2 + 1e-300)
        left = np.linalg.norm(dw - alpha * sw) / np.linalg.norm(dw)
        print("| %s | %d | %.1e | %.1e | %+.2f | %+.3g | %.2f |" % (name, ns, np.abs(s[~wake]).max(), np.abs(sw).max(), corr, alpha, left), flush=True)
        rec["variants"][name] = {"nsteps": ns, "shift": [[x.real, x.imag] for x in s], "corr": corr, "alpha": alpha, "left": left}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "wake_bisect.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
