#!/bin/bash
# GPU job 21: config 5 at full size on the new lx1 = 10 kernels (k_convect_mfma<10>, k_helm_p<10>, k_divgs_c3, k_schwarz_p): steps of 8,
# kernel timings, then a kernel trace of a 4-step map
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
SMOOTH=1 NPROJ=8 REPS=2 KERNELS="helm helm_wg convect convect_mfma divgs schwarz" timeout 900 python3 scripts/prof_cfg5.py 46 46 47 8 > $O/job21_steps.txt 2>&1
cat $O/job21_steps.txt
cd /tmp; export TMPDIR=/tmp
export SMOOTH=1 NPROJ=8 REPS=2
timeout 900 rocprofv3 --kernel-trace -d $O/p_c5t --output-format csv -- python3 $R/scripts/prof_cfg5.py 46 46 47 4 > $O/job21_trace_steps.txt 2> $O/job21.err
python3 $R/scripts/trace_summary.py $O/p_c5t --last 0.45 --min-calls 1 > $O/job21_cfg5_trace_summary.txt 2>&1
rm -rf $O/p_c5t
head -40 $O/job21_cfg5_trace_summary.txt
