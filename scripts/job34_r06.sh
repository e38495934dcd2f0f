#!/bin/bash
# GPU job 34: config 5 at full size, the time step of the FULL equations' map (Newton-Krylov) with the convection term on the matrix cores
# (k_convect_mfma_nl<10>) and with k_convect<10> (option mfma_convect = 0 by environment)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
MODE=nl SMOOTH=1 NPROJ=8 REPS=2 timeout 900 python3 scripts/prof_cfg5.py 46 46 47 8 > $O/job34_nl_mfma.txt 2>&1; tail -4 $O/job34_nl_mfma.txt
MFMA_CONVECT=0 MODE=nl SMOOTH=1 NPROJ=8 REPS=2 timeout 900 python3 scripts/prof_cfg5.py 46 46 47 8 > $O/job34_nl_thread.txt 2>&1; tail -4 $O/job34_nl_thread.txt
