#!/bin/bash
# GPU job 16: the lx1 = 10 E-apply kernels (config 5's pressure iteration) -- existing forms that were only ever timed at lx1 = 8
# (k_divgs_c3, k_schwarz_p) and the default forms forced to two workgroups per CU (experimental library)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
K="helm divgs divgs_c3 schwarz_wg schwarz_p"
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="$K" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 > $O/job16_default.txt 2>&1
cat $O/job16_default.txt
NSK_LIB=$R/nekstab_amd/lib/libnekstab_hip_exp.so SMOOTH=1 NPROJ=8 REPS=1 KERNELS="divgs schwarz_wg" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 > $O/job16_waves8.txt 2>&1
cat $O/job16_waves8.txt
