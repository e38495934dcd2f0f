#!/bin/bash
# Steady-state kernel trace of config 4 at full size (the r02 files were start-up traces): three maps of
# 40 time steps, summary over the LAST third of the time line only.  Run on the GPU box from the repository root.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out
export NSK_USE_GRAPH=0
NPROJ=32 rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg4 --output-format csv -- python3 $R/scripts/prof_cfg4.py 30 40 3 > $OUT/r03_cfg4_run.txt 2> $OUT/r03_cfg4_run.err
python3 $R/scripts/trace_summary.py $OUT/prof_cfg4 --last 0.3 > $OUT/r03_cfg4_trace_summary.txt
cat $OUT/r03_cfg4_run.txt >> $OUT/r03_cfg4_trace_summary.txt
rm -rf $OUT/prof_cfg4
