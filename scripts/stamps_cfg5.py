"""In-kernel phase stamps of the hexahedral k_helm at lx1 = 10 (config 5's kernel; a 24^3 box fills the chip 54 times over).
Needs the -DNSK_STAMPS build (scripts/stamps3d.py).   python scripts/stamps_cfg5.py [n=24]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import capi, mesh3d
capi.LIB_PATH = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip_stamps.so")
from nekstab_amd.capi import NekStabHip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
stretch = lambda xi: 0.5 * (1.0 - np.cos(np.pi * xi))
c = mesh3d.box_case_3d(n, n, n, 10, lengths=(1.0, 1.0, 1.0), re=1000.0, endtime=0.02, stretch=stretch)
sx, sy, sz = np.sin(np.pi * c.x), np.sin(np.pi * c.y), np.sin(np.pi * c.z)
c.ub[0] = sx ** 2 * np.sin(2 * np.pi * c.y) * sz ** 2 * c.mask
c.ub[1] = -np.sin(2 * np.pi * c.x) * sy ** 2 * sz ** 2 * c.mask
h = NekStabHip(c, c.meta["vert"], c.meta["nvert"], tol_helm=1e-9, tol_pres=1e-2, tol_relative=1, max_helm_iter=400, max_pres_iter=192, nproj=0)
q, f = h.alloc(2)
w = 1e-2 * np.sin(2 * np.pi * c.x) * np.sin(3 * np.pi * c.y) * np.sin(2 * np.pi * c.z) * c.mask
h.upload3(q, c.ub[0] + w, c.ub[1] - w, w, np.zeros(h.npres))
h.set_nsteps(2)
h.matvec(f, q, 0)
nb = c.nel
names = ["CG scalars (+ table, corner entries, fragments in the same trip)", "loads issued; corner values of wavefront 0 arrived", "barrier", "updates of three components + stores issued", "barrier", "A z of three components (matrix cores) + stores + wave sums"]
ns = len(names) + 1
out = np.zeros(16 * nb, dtype=np.uint64)
fn = h.lib.nsk_debug_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = fn(h.ctx, out.ctypes.data, -nb); assert rc == 0
t = out.reshape(nb, 16)[:, :ns].astype(np.int64)
t = t[t[:, 0] > 0]
rel = (t - t[:, 0].min()) * 0.01
print("k_helm<10>: %d workgroups, starts spread over %.1f us, last stamp at %.1f us" % (len(t), rel[:, 0].max(), rel[:, ns - 1].max()))
d = np.diff(rel, axis=1)
for i, name in enumerate(names):
    print("  %-66s median %6.2f us   p10 %6.2f   p90 %6.2f" % (name, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
life = rel[:, ns - 1] - rel[:, 0]
print("  workgroup lifetime median %.2f us; workgroups in flight = %.1f per CU" % (np.median(life), life.sum() / rel[:, ns - 1].max() / 256))
