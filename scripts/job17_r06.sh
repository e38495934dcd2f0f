#!/bin/bash
# GPU job 17: lx1 = 10, the wavefront-per-element Schwarz form that exists (k_schwarz_w<10>) next to the others
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="${K:-schwarz_wg schwarz_p schwarz_w schwarz_w16 divgs_w divgs_c3}" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 > $O/job17.txt 2>&1
cat $O/job17.txt
