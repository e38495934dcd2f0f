import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import capi, mesh
capi.LIB_PATH = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip_stamps.so")
from nekstab_amd.capi import NekStabHip
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, nproj=8)
rng = np.random.default_rng(0)
vq, vf = h.alloc(2)
h.upload(vq, rng.standard_normal(case.x.shape) * case.mask, rng.standard_normal(case.x.shape) * case.mask, np.zeros(h.npres))
h.set_nsteps(5); h.matvec(vf, vq, 0)
nb = 499
out = np.zeros(16 * nb, dtype=np.uint64)
fn = h.lib.nsk_debug_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = fn(h.ctx, out.ctypes.data, nb); assert rc == 0
t = out.reshape(nb, 16)[:, :5].astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) * 0.01     # us (100 MHz)
print("block start spread (us): min %.2f med %.2f max %.2f" % (rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max()))
d = np.diff(rel, axis=1)
for i, name in enumerate(["done-flag load", "loads+gs gather -> LDS", "opdiv (2 barriers)", "dots + partials"]):
    print("%-28s med %.2f us  p90 %.2f" % (name, np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("kernel end (last stamp) max %.2f us" % rel[:, 4].max())

out = np.zeros(16 * nb, dtype=np.uint64)
rc = fn(h.ctx, out.ctypes.data, -nb); assert rc == 0
t = out.reshape(nb, 16)[:, :8].astype(np.int64)
rel = (t - t[:, 0].min()) * 0.01
print("k_helm (it=5): block start spread max %.2f us" % rel[:, 0].max())
d = np.diff(rel, axis=1)
for i, name in enumerate(["flag check", "issue phase-A loads", "issue gs loads (needs table)", "partials reduce -> alpha,beta", "updates + LDS write (waits loads)", "axhelm", "block_reduce<8>", "store partials"]):
    if i < 7: print("%-36s med %.2f us  p90 %.2f" % (name, np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("kernel end max %.2f us" % rel[:, 7].max())
