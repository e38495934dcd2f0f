#!/usr/bin/env python3
"""Maps per second of BASELINE config 2 with one map at a time and with b maps in flight on b lanes (nsk_matvec_batch), inside the
factorisations that use them: the single-vector Arnoldi (the pinned default) and the band Arnoldi with b seeds, same number of
maps, production settings.     python scripts/lanes_bench.py [k] [b ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import krylov, mesh, seed
from nekstab_amd.settings import production_context
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bs = [int(x) for x in sys.argv[2:]] or [2, 3]
MODE = int(os.environ.get("MODE", "0"))
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8, adjoint=bool(MODE))
qx, qy = seed.add_noise(case)
rng = np.random.default_rng(0)
h = production_context(case)
s0 = h.alloc(1)[0]
h.upload(s0, qx, qy, np.zeros(h.npres))
t0 = time.perf_counter(); r1 = krylov.krylov_schur(h, s0, k, mode=MODE, schur_tgt=0); w1 = time.perf_counter() - t0
st = h.stats()
print("single vector: %d maps in %.2f s = %.2f maps/s (%.2f + %.2f iterations per step); leading Ritz value %.8f%+.8fi residual %.1e" % (
    k, w1, k / w1, st["total_helm_iters"] / st["total_steps"], st["total_pres_iters"] / st["total_steps"], r1.vals[0].real, abs(r1.vals[0].imag), r1.residual[0]), flush=True)
h.free(r1.Q)
for b in bs:
    seeds = h.alloc(b)
    h.copy(seeds[0], s0)
    for j in range(1, b):
        # further seeds: the noise field modulated by smooth functions of the coordinates (continuous, admissible, independent)
        h.upload(seeds[j], qy * np.cos(0.2 * j * case.x), qx * np.cos(0.3 * j * case.y), np.zeros(h.npres))
    t0 = time.perf_counter(); rb = krylov.band_arnoldi(h, seeds, k, mode=MODE); wb = time.perf_counter() - t0
    print("band width %d: %d maps in %.2f s = %.2f maps/s (x%.2f); leading Ritz value %.8f%+.8fi residual %.1e" % (
        b, k, wb, k / wb, (k / wb) / (k / w1), rb.vals[0].real, abs(rb.vals[0].imag), rb.residual[0]), flush=True)
    for l in range(b):
        ls = h.lane_stats(l)
        print("     lane %d: %d steps, %.2f + %.2f iterations per step, %d graph re-captures, %d redone maps" % (l, ls["total_steps"], ls["total_helm_iters"] / max(ls["total_steps"], 1), ls["total_pres_iters"] / max(ls["total_steps"], 1), ls["recaptures"], ls["retries"]))
    h.free(rb.Q); h.free(seeds)
h.close()
