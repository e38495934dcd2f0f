#!/bin/bash
# GPU job 25: the hexahedral tests on the new lx1 = 10 defaults, then config 5 at full size (steps of 8, kernel timings)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_3d_gpu.py -q -x 2>&1 | tail -4
SMOOTH=1 NPROJ=8 REPS=2 KERNELS="helm helm_wg convect_mfma divgs divgs_wg schwarz schwarz_p schwarz_wg" timeout 900 python3 scripts/prof_cfg5.py 46 46 47 8 > $O/job25_steps.txt 2>&1
cat $O/job25_steps.txt
