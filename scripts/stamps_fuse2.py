"""In-kernel phase stamps of the two-launch GMRES iteration (k_schwarz_uc: Schwarz role and coarse role; k_divgs_t) on config 2.
Needs the -DNSK_STAMPS build (scripts/stamps3d.py has the command).   python3 scripts/stamps_fuse2.py [j=3]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import capi, mesh, seed
capi.LIB_PATH = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip_stamps.so")
from nekstab_amd.settings import production_context
jd = sys.argv[1] if len(sys.argv) > 1 else "3"
os.environ["NSK_STAMP_J"] = jd
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), 8)
h = production_context(case)
vq, vf = h.alloc(2)
qx, qy = seed.add_noise(case)
h.upload(vq, qx, qy, np.zeros(h.npres))
h.set_nsteps(8); h.matvec(vf, vq, 0)
nb, ncg = 499, 255
fn = h.lib.nsk_debug_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]


def grab(kernel, nrow):
    os.environ["NSK_STAMP_KERNEL"] = kernel
    out = np.zeros(16 * nrow, dtype=np.uint64)
    rc = fn(h.ctx, out.ctypes.data, nrow); assert rc == 0, rc
    return out.reshape(nrow, 16).astype(np.int64)


def report(title, t, names):
    ok = t[:, 0] > 0
    t = t[ok]
    t0 = t[:, 0].min()
    rel = (t[:, :len(names) + 1] - t0) * 0.01
    print("%s (%d workgroups): start spread med %.2f max %.2f us, END med %.2f max %.2f us" % (title, len(t), np.median(rel[:, 0]), rel[:, 0].max(), np.median(rel[:, len(names)]), rel[:, len(names)].max()))
    d = np.diff(rel, axis=1)
    for i, nm in enumerate(names):
        print("    %-58s med %5.2f  p90 %5.2f us" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))


t = grab("schwarz_uc", nb + ncg)
report("k_schwarz_uc j=%s, Schwarz role" % jd, t[:nb], ["done flag, patch indices issued", "all loads issued (needs the patch indices), basis -> LDS", "partial sums + barrier",
                                                        "column rotation (one lane) + barrier", "v_j, store, patch -> LDS + barrier", "patch solve", "D^T (2 passes) + stores"])
report("k_schwarz_uc j=%s, coarse role" % jd, t[nb:nb + ncg], ["done flag", "loads + vertex sums -> LDS (3 chunks)", "partial sums + barrier", "column rotation (lane 0)", "matrix product + wave sums", "barrier (column rotated)", "combine with the history, store"])
t = grab("divgs_t", nb)
report("k_divgs_t j=%s" % jd, t[:nb], ["done flag", "first trip issued (table, metrics, Tc, V_k, Z)", "second trip issued (gathers, x_c) + sums -> LDS", "barrier", "divergence + Tc x_c + stores", "restriction + dots + partials"])
