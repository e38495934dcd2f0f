#!/bin/bash
# GPU job 19: k_convect_mfma<10> -- parity (tests/test_3d_gpu.py at lx1 = 10) and timing next to k_convect<10> at 24^3 elements
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_3d_gpu.py -q -x -k "lx10 or forms_agree or kernel_forms" 2>&1 | tail -5
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="convect convect_mfma schwarz divgs" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 > $O/job19.txt 2>&1
cat $O/job19.txt
