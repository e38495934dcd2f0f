#!/bin/bash
# Steady-state kernel trace of config 4 at full size (the r02 files were start-up traces): ten maps of
# 40 time steps -- the launch budgets of ALL step classes are cut to the window maximum only once eight maps are on record
# (budgets_update) -- and the summary covers the LAST map only (the last 9 % of the time line).  Run on the GPU box from the repository root.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out
export NSK_USE_GRAPH=0
NPROJ=32 rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg4 --output-format csv -- python3 $R/scripts/prof_cfg4.py 30 40 10 > $OUT/r03_cfg4_run.txt 2> $OUT/r03_cfg4_run.err
python3 $R/scripts/trace_summary.py $OUT/prof_cfg4 --last 0.09 > $OUT/r03_cfg4_trace_summary.txt
cat $OUT/r03_cfg4_run.txt >> $OUT/r03_cfg4_trace_summary.txt
rm -rf $OUT/prof_cfg4
