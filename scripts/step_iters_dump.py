#!/usr/bin/env python3
"""Per-time-step CG / GMRES iteration counts of every map of an Arnoldi run at the production settings (nsk_get_step_iters):
the data the per-step launch budgets were designed on.   python3 scripts/step_iters_dump.py [k=64] [out.npz]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from nekstab_amd import mesh, seed
from nekstab_amd.settings import production_context

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r05", "step_iters.npz")
case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
h = production_context(case)
qx, qy = seed.add_noise(case)
Q = h.alloc(k + 1)
h.upload(Q[0], qx, qy, np.zeros(h.npres))
h.scal(Q[0], 1.0 / h.norm(Q[0]))
HH, PP = [], []
for j in range(k):
    h.matvec(Q[j + 1], Q[j], 0)
    hh, pp = h.step_iters()
    HH.append(hh.copy()); PP.append(pp.copy())
    h.orth(Q[j + 1], Q[:j + 1])
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, helm=np.array(HH), pres=np.array(PP))
HH, PP = np.array(HH), np.array(PP)
print("maps", HH.shape, "helm mean %.2f pres mean %.2f" % (HH.mean(), PP.mean()), "retries", h.stats()["retries"])
print("helm, last map:", HH[-1].tolist())
print("pres, last map:", PP[-1].tolist())
h.close()
