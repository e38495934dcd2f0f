"""In-kernel phase stamps (s_memrealtime, 100 MHz) of the hexahedral pressure kernels at config 4's size.
Needs the -DNSK_STAMPS build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DNSK_STAMPS -o nekstab_amd/lib/libnekstab_hip_stamps.so nekstab_amd/csrc/nsk.hip

    NSK_STAMP_KERNEL=schwarz|schwarz_p|schwarz_w|divgs|helm [NSK_STAMP_J=5] python scripts/stamps3d.py [nz=30]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import capi, mesh, mesh3d
capi.LIB_PATH = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip_stamps.so")
from nekstab_amd.capi import NekStabHip
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = os.path.join(ROOT, "tests", "golden")
c2 = mesh.load_case_npz(os.path.join(G, "backstep_case.npz"), 8, re=500.0, endtime=1.0, xlspg=5.0, xrspg=10.0, spng_str=2.0)
c3 = mesh3d.extrude_case(c2, nz, 0.2 * nz, periodic=True)
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-10, tol_pres=1e-2, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=0)
rng = np.random.default_rng(0)
q, f = h.alloc(2)
h.upload3(q, *(rng.standard_normal(c3.x.shape) * c3.mask for _ in range(3)), np.zeros(h.npres))
h.set_nsteps(2)
try:
    h.matvec(f, q, 1)          # (only to leave realistic data in the solver arrays; a noise seed may hit the iteration cap)
except capi.NskError as e:
    print("note:", e)
nb = c3.nel
which = os.environ.get("NSK_STAMP_KERNEL", "divgs")
names = {"divgs": ["loads + dssum gather -> LDS tile", "barrier", "opdiv3 (MFMA passes)", "dots + partials"],
         "schwarz_w": ["loads -> LDS", "forward passes", "backward passes", "restrict + coarse term", "D^T (3 components) + stores"],
         "schwarz_p": ["barrier, coarse term, metric loads issued", "forward passes", "metrics -> LDS (wait)", "prefetch issue + backward passes", "restrict + products", "opgradt3", "next tile -> LDS (wait)", "stores"],
         "helm": ["CG scalars (htot, hscal)", "loads issued; corner values of wavefront 0 arrived", "barrier", "updates of three components + stores issued", "barrier", "A z of three components (matrix cores) + stores + wave sums"],
         "schwarz": ["loads: factors, metrics, patch gather", "barrier", "fast diagonalisation, 6 passes", "restrict + opgradt3 (MFMA passes)", "stores"]}[which]
ns = len(names) + 1
out = np.zeros(16 * nb, dtype=np.uint64)
fn = h.lib.nsk_debug_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = fn(h.ctx, out.ctypes.data, -nb if which == "helm" else nb); assert rc == 0
t = out.reshape(nb, 16)[:, :ns].astype(np.int64)
ok = t[:, 0] > 0
t = t[ok]
rel = (t - t[:, 0].min()) * 0.01     # us
print("%s: %d workgroups, last launch of five: starts spread over %.1f us, last stamp at %.1f us" % (which, len(t), rel[:, 0].max(), rel[:, ns - 1].max()))
d = np.diff(rel, axis=1)
for i, name in enumerate(names):
    print("  %-42s median %6.2f us   p10 %6.2f   p90 %6.2f" % (name, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
life = rel[:, ns - 1] - rel[:, 0]
print("  workgroup lifetime median %.2f us; workgroups in flight (sum of lifetimes / kernel span) = %.0f = %.1f per CU" % (np.median(life), life.sum() / rel[:, ns - 1].max(), life.sum() / rel[:, ns - 1].max() / 256))
