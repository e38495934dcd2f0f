#!/bin/bash
# GPU job 33: k_convect_mfma_nl<10> (the full equations' convection term on the matrix cores) -- parity, timing at 24^3 elements
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_3d_gpu.py -q -x -k "convection" > $O/job33.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/job33.txt
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="convect_nl convect_mfma_nl convect_mfma" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | tail -5
