"""Developer probe: full matvec on the GPU vs the oracle, timings and solver statistics."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with_oracle = int(sys.argv[2]) if len(sys.argv) > 2 else 1
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
case = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
modes = np.load(os.path.join(ROOT, "tests/golden/cylinder_modes.npz"))
t0 = time.time()
nproj = int(sys.argv[4]) if len(sys.argv) > 4 else 0
h = NekStabHip(case, case.meta["vert"], case.meta["nvert"], schwarz_layers=layers, max_helm_iter=100, max_pres_iter=48, nproj=nproj)
print("init %.2fs dt=%g nsteps=%d" % (time.time() - t0, h.dt, h.nsteps))
rng = np.random.default_rng(0)
if lx1 == 6:
    u = modes["dRe_u"].astype(np.float64); p1 = modes["dRe_p"].astype(np.float64)
else:
    u = mesh.interp_field_2d(modes["dRe_u"].astype(np.float64), lx1); p1 = mesh.interp_field_2d(modes["dRe_p"].astype(np.float64), lx1)
from nekstab_amd.quadrature import gauss_lobatto_legendre, gauss_legendre, interp_matrix
J = interp_matrix(gauss_lobatto_legendre(lx1)[0], gauss_legendre(lx1 - 2)[0])
q = (u[0], u[1], J @ p1 @ J.T)
vq, vf = h.alloc(2)
h.upload(vq, *q)
res = {}
import itertools
cfgs = {"tight": (1e-13, 1e-13, 0)}
for th, tp in itertools.product((1e-11,), (0.5, 0.2, 0.1, 0.03, 0.01)):
    cfgs["rel h%g p%g" % (th, tp)] = (th, tp, 1)
for name, (th, tp, rel) in cfgs.items():
    h.set_tolerances(th, tp, rel)
    times = []
    for rep in range(3):
        t0 = time.time()
        try:
            h.matvec(vf, vq, 0)
        except Exception as e:
            print(name, "ERR", e)
        times.append(time.time() - t0)
    res[name] = h.download(vf)
    st = h.stats()
    print("%-16s matvec %s  helm/step %.1f pres/step %.1f max %d/%d budget %d/%d" % (name, " ".join("%.3f" % t for t in times),
          st["helm_iters"] / h.nsteps, st["pres_iters"] / h.nsteps, st["max_helm_iter"], st["max_pres_iter"], st["budget_helm"], st["budget_pres"]))
w = None
def relv(a, b):
    return np.sqrt(sum(np.sum((x - y) ** 2) for x, y in zip(a[:2], b[:2])) / sum(np.sum(y ** 2) for y in b[:2]))
for k in res:
    print(k, "vs tight: vel", relv(res[k], res["tight"]), "pres", np.abs(res[k][2] - res["tight"][2]).max() / np.abs(res["tight"][2]).max())
if with_oracle:
    from tests.conftest import make_oracle
    o = make_oracle(case)
    t0 = time.time(); ref = o.matvec(q); print("oracle matvec %.1fs" % (time.time() - t0))
    for k in res:
        print(k, "vs oracle: vel", relv(res[k], ref), "pres", np.abs(res[k][2] - ref[2]).max() / np.abs(ref[2]).max())
