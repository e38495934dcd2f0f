"""Ablation of the hexahedral k_helm (developer switches: 1 = no axhelm, 2 = no gather)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nekstab_amd import mesh, mesh3d
from nekstab_amd.capi import NekStabHip
lx1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
c2 = mesh.load_case_npz(os.path.join(ROOT, "tests/golden/cylinder_case.npz"), lx1)
c3 = mesh3d.extrude_case(c2, 2, 1.0, periodic=True)
h = NekStabHip(c3, c3.meta["vert"], c3.meta["nvert"], tol_helm=1e-11, tol_pres=1e-1, tol_relative=1, max_helm_iter=150, max_pres_iter=48, nproj=8)
rng = np.random.default_rng(0)
v0, v1 = h.alloc(2)
h.upload3(v0, *(rng.standard_normal(c3.x.shape) * c3.mask for _ in range(3)), np.zeros(h.npres))
h.set_nsteps(2); h.matvec(v1, v0, 0)
for dbg in (0, 1, 2, 3):
    h.set_option("dbg", dbg)
    print("dbg", dbg, h.bench_kernel("helm", 100), flush=True)
