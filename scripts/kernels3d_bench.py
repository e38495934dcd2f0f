"""HIP-event timings of the hexahedral hot kernels at config 4's size (nsk_bench_kernel), with the algorithmic bytes of
nekstab_amd/roofline.py next to them.   python scripts/kernels3d_bench.py [nz=30] [kernel ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nekstab_amd import mesh, mesh3d, capi
from nekstab_amd.capi import NekStabHip
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 30
names = sys.argv[2:] or ["helm", "divgs", "schwarz", "schwarz_wg", "gs_dots3", "gs_lag3", "gs_dots8", "gs_lag8", "gs_dots16", "gs_lag16", "gs_dots32", "gs_lag32", "pres_update", "vel_update_proj", "pres_rhs", "rhs", "convect_mfma"]
G = os.path.join(ROOT, "tests", "golden")
c2 = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
c3 = mesh3d.extrude_case(c2, nz, 0.2 * nz, periodic=True)
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=int(os.environ.get("NPROJ", "32")))
tg = np.load(os.path.join(G, "backstep_tg.npz"))
u2 = mesh.interp_field_2d(tg["pRe_u"].astype(np.float64), 8) * c2.mask
w = 1e-2 * np.sin(2 * np.pi * c3.z / (0.2 * nz)) * c3.mask * np.abs(mesh3d.extrude_field(u2[0], nz))
q, f = h.alloc(2)
h.upload3(q, mesh3d.extrude_field(u2[0], nz), mesh3d.extrude_field(u2[1], nz), w, np.zeros(h.npres))
h.scal(q, 1.0 / h.norm(q))
h.set_nsteps(3)
try:
    h.matvec(f, q, 1)          # (only to leave realistic data in the solver arrays)
except capi.NskError as exc:
    print("note:", exc)
P, P2 = h.nvel, h.npres
print("E = %d, P = %.2f M, P2 = %.2f M" % (c3.nel, 1e-6 * P, 1e-6 * P2))
if os.environ.get("ZERO_METRICS") is not None:
    h.set_option("zero_metrics", int(os.environ["ZERO_METRICS"]))
print("zero_arrays = 0x%x" % h.stats().get("zero_arrays", 0))
for n in names:
    try:
        r = h.bench_kernel(n, int(os.environ.get("REPS", "20")))
        print("%-18s %9.1f us" % (n, r["avg_us"]), flush=True)
    except capi.NskError as e:
        print("%-18s %s" % (n, e))
