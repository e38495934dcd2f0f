#!/bin/bash
# GPU job 18: config 5 at full size, kernel trace of a map with the lx1 = 10 E-apply forms that job 16 found faster
# (k_divgs_c3, k_schwarz_p; by environment on the committed library) -- where does the time step go?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export NSK_EAPPLY_PIPE=${PIPE:-1} NSK_DIVGS_C3=${C3:-1} SMOOTH=1 NPROJ=8 REPS=2
timeout 900 rocprofv3 --kernel-trace -d $O/p_c5t --output-format csv -- python3 $R/scripts/prof_cfg5.py 46 46 47 4 > $O/job18_steps.txt 2> $O/job18.err
python3 $R/scripts/trace_summary.py $O/p_c5t --last 0.45 --min-calls 1 > $O/job18_cfg5_trace_summary.txt 2>&1
rm -rf $O/p_c5t
cat $O/job18_steps.txt; head -60 $O/job18_cfg5_trace_summary.txt
