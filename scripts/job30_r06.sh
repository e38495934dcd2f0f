#!/bin/bash
# GPU job 30: k_divgs_q (small workgroups) -- parity with the other forms, timings at config 4's size (lx1 = 8) and at 24^3 elements of lx1 = 10
# (ran on an experimental build: k_divgs_q and its bench names are NOT in the tree -- DESIGN.md section 7)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python3 -m pytest tests/test_3d_gpu.py -q -x -k "forms_agree" 2>&1 | tail -4
REPS=20 timeout 900 python3 scripts/kernels3d_bench.py 30 divgs divgs_q2 divgs_q4 divgs_c3 2>&1 | tail -5
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="divgs divgs_q4 divgs_wg" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | tail -5
