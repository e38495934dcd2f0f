#!/usr/bin/env python3
"""Row-by-row comparison of a k_dim = 200 Arnoldi spectrum with the reference's Spectre_H{d,a}.dat under several
inner-solver settings.  Output: one table per (case, setting) on stdout and
gpurun_out/spectrum_pin.json.

    python scripts/spectrum_pin.py [--cases d6,a8] [--settings tight,bench,nek]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = {
    # name: (tol_helm, tol_pres, relative, nproj, options)
    "tight": (1e-11, 1e-2, 1, 0, {}),
    "tighter": (1e-13, 1e-4, 1, 0, {}),
    "bench_r1": (1e-9, 3e-1, 1, 8, {"min_pres_iter": 2, "pres_cap": 4}),
    "nocap": (1e-9, 3e-1, 1, 8, {"min_pres_iter": 2}),
    "mid": (1e-10, 1e-1, 1, 8, {"min_pres_iter": 2}),
    "nek_abs": (1e-9, 1e-7, 0, 8, {"min_pres_iter": 1}),
    "tighter2": (1e-13, 1e-6, 1, 0, {}),
    "mid_np": (1e-10, 1e-1, 1, 0, {"min_pres_iter": 2}),
    "t10_2": (1e-10, 1e-2, 1, 0, {"min_pres_iter": 2}),
    "t10_2p": (1e-10, 1e-2, 1, 8, {"min_pres_iter": 2}),
    "t9_1np": (1e-9, 1e-1, 1, 0, {"min_pres_iter": 2}),
    "m1": (1e-10, 1e-1, 1, 8, {"min_pres_iter": 2, "pres_cap": 6}),
    "m2": (1e-10, 2e-1, 1, 8, {"min_pres_iter": 2}),
    "m3": (1e-10, 3e-1, 1, 8, {"min_pres_iter": 2}),
    "m4": (1e-10, 1e-1, 1, 8, {"min_pres_iter": 3}),
    "m5": (1e-11, 1e-1, 1, 8, {"min_pres_iter": 2}),
    "m6": (1e-10, 1e-1, 1, 8, {"min_pres_iter": 2, "early_pres_mul": 1.0}),
    "m7": (1e-10, 5e-2, 1, 8, {"min_pres_iter": 2}),
    "m8": (1e-11, 3e-2, 1, 8, {"min_pres_iter": 2}),
    "m9": (1e-12, 1e-3, 1, 8, {"min_pres_iter": 2}),
    "h12p1": (1e-12, 1e-1, 1, 8, {"min_pres_iter": 2}),
    "h12p2": (1e-12, 2e-1, 1, 8, {"min_pres_iter": 2}),
    "h12p3": (1e-12, 3e-1, 1, 8, {"min_pres_iter": 2}),
    "h13p1": (1e-13, 1e-1, 1, 8, {"min_pres_iter": 2}),
    "h12p1c6": (1e-12, 1e-1, 1, 8, {"min_pres_iter": 2, "pres_cap": 6}),
    "h12p1np": (1e-12, 1e-1, 1, 0, {"min_pres_iter": 2}),
    "h11p1": (1e-11, 1e-1, 1, 8, {"min_pres_iter": 2}),
    "h11p1n16": (1e-11, 1e-1, 1, 16, {"min_pres_iter": 2}),
    "h11p1n24": (1e-11, 1e-1, 1, 24, {"min_pres_iter": 2}),
    "h11p1n32": (1e-11, 1e-1, 1, 32, {"min_pres_iter": 2}),
    "h11p2n20": (1e-11, 2e-1, 1, 20, {"min_pres_iter": 2}),
    "h11p1n20m1": (1e-11, 1e-1, 1, 20, {"min_pres_iter": 1}),
    "h12p1n20": (1e-12, 1e-1, 1, 20, {"min_pres_iter": 2}),
}


def run(case, mode, table, name, k_dim):
    from nekstab_amd import krylov, seed
    from nekstab_amd.capi import NekStabHip
    th, tp, rel, nproj, opts = SETTINGS[name]
    h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=th, tol_pres=tp, tol_relative=rel,
                   schwarz_layers=2, max_helm_iter=120, max_pres_iter=48, nproj=nproj)
    for k, v in opts.items():
        h.set_option(k, v)
    qx, qy = seed.add_noise(case)
    v0, v1 = h.alloc(2)
    h.upload(v0, qx, qy, np.zeros(h.npres))
    h.scal(v0, 1.0 / h.norm(v0))
    h.matvec(v1, v0, mode)
    t0 = time.perf_counter()
    res = krylov.krylov_schur(h, v1, k_dim, mode=mode, schur_tgt=0)
    wall = time.perf_counter() - t0
    rows = []
    ref = table
    seen = set()
    for n, r in enumerate(ref):
        if r[2] >= 1e-6 or r[1] < 0:
            continue
        z = complex(r[0], r[1])
        j = int(np.argmin(np.abs(res.vals - z)))
        rows.append({"row": n + 1, "ref": [float(r[0]), float(r[1])], "ref_residual": float(r[2]),
                     "ours": [float(res.vals[j].real), float(res.vals[j].imag)], "residual": float(res.residual[j]),
                     "diff": float(abs(res.vals[j] - z))})
    st = h.stats()
    h.close()
    print("   helm/step %.2f pres/step %.2f  matvecs/s %.2f" % (st["total_helm_iters"] / max(st["total_steps"], 1), st["total_pres_iters"] / max(st["total_steps"], 1), k_dim / wall))
    return {"setting": name, "wall_s": wall, "helm_per_step": st["total_helm_iters"] / max(st["total_steps"], 1), "pres_per_step": st["total_pres_iters"] / max(st["total_steps"], 1), "rows": rows, "retries": st["retries"], "vals": [[float(v.real), float(v.imag)] for v in res.vals[:60]],
            "resid": [float(x) for x in res.residual[:60]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="d6,a8")
    ap.add_argument("--settings", default="tight,bench_r1,nocap,mid,nek_abs")
    ap.add_argument("--kdim", type=int, default=200)
    a = ap.parse_args()
    from nekstab_amd import mesh
    spectre = np.load(os.path.join(ROOT, "tests", "golden", "cylinder_spectre.npz"))
    out = {}
    for cs in a.cases.split(","):
        adj = cs[0] == "a"
        lx1 = int(cs[1:])
        case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), lx1, adjoint=adj)
        table = spectre["Ha" if adj else "Hd"]
        for name in a.settings.split(","):
            try:
                r = run(case, 1 if adj else 0, table, name, a.kdim)
            except Exception as e:      # a setting that diverges is a result too
                print("%s %s FAILED: %s" % (cs, name, e), flush=True)
                out["%s/%s" % (cs, name)] = {"error": str(e)}
                continue
            out["%s/%s" % (cs, name)] = r
            print("== %s  %s  wall %.1fs retries %d" % (cs, name, r["wall_s"], r["retries"]))
            for row in r["rows"]:
                print("  row %2d ref %.7f%+.7fi (%.1e)  ours %.9f%+.9fi (%.1e)  diff %.2e" % (
                    row["row"], row["ref"][0], row["ref"][1], row["ref_residual"], row["ours"][0], row["ours"][1], row["residual"], row["diff"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "spectrum_pin.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
