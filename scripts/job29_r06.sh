#!/bin/bash
# GPU job 29: config 4's size (50 100 hexahedra, lx1 = 8): the E-apply forms as they stand after round 5's load restructuring
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
REPS=20 timeout 900 python3 scripts/kernels3d_bench.py 30 divgs divgs_c3 divgs_w schwarz schwarz_p helm convect_mfma convect 2>&1 | tail -12
