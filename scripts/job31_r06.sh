#!/bin/bash
# GPU job 31: lx1 = 8 convection on the matrix cores -- k_convect_mfma8 (seven LDS tiles, 75 KB, two workgroups per CU) against the generic
# k_convect_mfma<8> (three regions, 41 KB, 80 registers: three per CU); parity of the generic form through the test hook
# (ran on an experimental build: option mfma_convect = 2 / bench name convect_mfma_g are NOT in the tree -- DESIGN.md section 7)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from tests.test_3d_gpu import _case, _hip, _rel
c = _case(8, True); h = _hip(c)
rng = np.random.default_rng(2); u = rng.standard_normal((3,) + c.x.shape)
a0, a1 = h.t_op3(8, u, 0), h.t_op3(8, u, 1)
h.set_option("mfma_convect", 2)
b0, b1 = h.t_op3(8, u, 0), h.t_op3(8, u, 1)
print("generic against k_convect_mfma8: direct %.2e adjoint %.2e; against the thread-per-node kernel %.2e" % (_rel(b0, a0), _rel(b1, a1), _rel(b0, h.t_op3(3, u, 0))))
h.close()
PY
REPS=20 timeout 900 python3 scripts/kernels3d_bench.py 30 convect_mfma convect_mfma_g convect_mfma convect_mfma_g 2>&1 | tail -5
