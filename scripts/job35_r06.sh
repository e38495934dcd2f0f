#!/bin/bash
# GPU job 35: the four-wavefront Schwarz form (k_schwarz_q) at lx1 = 8 against the sixteen-per-CU wavefront form, config 4's size
# (ran with launch_schwarz3 patched to take form 5 at lx1 = 8 under NSK_SCHWQ_ALL: NOT in the tree -- DESIGN.md section 7)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
NSK_SCHWQ_ALL=1 REPS=20 timeout 900 python3 scripts/kernels3d_bench.py 30 schwarz schwarz_q schwarz schwarz_q 2>&1 | tail -5
