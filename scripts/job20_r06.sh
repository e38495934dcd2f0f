#!/bin/bash
# GPU job 20: k_helm_p<10> (resident workgroups, LDS-DMA prefetch) -- bit identity against k_helm<10>, then timings at 24^3 elements
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_3d_gpu.py -q -x -k "resident_helmholtz or lx10" 2>&1 | tail -8
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="helm helm_wg" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 > $O/job20.txt 2>&1
cat $O/job20.txt
NSK_HELM_PF=0 SMOOTH=1 NPROJ=8 REPS=1 timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | tail -3
