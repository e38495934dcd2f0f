"""How well do given inner-solver settings pin the direct spectrum, and how reproducible is that?  For every setting: k_dim = 200
Arnoldi runs of the Re = 50 cylinder (direct) in two arithmetically equivalent realisations (classic / merged GMRES bookkeeping:
one matvec equal to 2e-11), each compared row by row with the converged spectrum in tests/golden/cylinder_converged_spectra.npz.

    python scripts/pin_noise.py [lx1] [setting ...]      setting = tol_helm:tol_pres:nproj[:min_pres_iter]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sets = sys.argv[2:] or ["1e-11:1e-1:16", "1e-11:3e-2:16", "1e-12:3e-2:16", "1e-12:1e-2:16", "3e-12:1e-2:16"]
conv = np.load(os.path.join(ROOT, "tests", "golden", "cylinder_converged_spectra.npz"))["Hd%d" % lx1]
rows = [complex(r[0], r[1]) for r in conv if r[2] < 1e-8 and r[1] >= 0]
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), lx1)
print("lx1 %d, %d converged rows: %s" % (lx1, len(rows), " ".join("%.4f%+.4fi" % (z.real, z.imag) for z in rows)), flush=True)
for sname in sets:
    f = sname.split(":")
    th, tp, npj = float(f[0]), float(f[1]), int(f[2])
    mp = int(f[3]) if len(f) > 3 else 2
    for merged in (1, 0):
        h = production_context(case, tol_helm=th, tol_pres=tp, nproj=npj, min_pres_iter=mp)
        h.set_option("merged_update", merged)
        qx, qy = seed.add_noise(case)
        v0, v1 = h.alloc(2)
        h.upload(v0, qx, qy, np.zeros(h.npres))
        h.scal(v0, 1.0 / h.norm(v0))
        h.matvec(v1, v0, 0)
        t0 = time.perf_counter()
        res = krylov.krylov_schur(h, v1, 200, mode=0, schur_tgt=0)
        wall = time.perf_counter() - t0
        st = h.stats()
        d = [abs(res.vals[int(np.argmin(np.abs(res.vals - z)))] - z) for z in rows]
        print("%-18s merged %d: %.2f matvecs/s, %.2f + %.2f iterations/step | worst %.1e | %s" % (
            sname, merged, 200 / wall, st["total_helm_iters"] / max(st["total_steps"], 1), st["total_pres_iters"] / max(st["total_steps"], 1),
            max(d), " ".join("%.0e" % x for x in d)), flush=True)
        h.close()
