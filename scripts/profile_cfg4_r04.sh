#!/bin/bash
# Round-4 measurements of config 4 at full size (E = 50 100 hexahedra, lx1 = 8, adjoint map, host-checked eager steps):
#   1. time per step with the classic Gram-Schmidt sequence (NSK_GS_LAG=0), the lagged one (1), and lagged + separate first-pass dots (2)
#   2. kernel trace of the default build (steady state: the last map of four)
#   3. FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel trace only) -> per-kernel HBM-side bytes
# Run on the GPU box from the repository root; summaries land in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out
TAG=${TAG:-r04}
for lag in 0 1 2; do
  NSK_GS_LAG=$lag NPROJ=32 python3 $R/scripts/prof_cfg4.py 30 40 3 > $OUT/${TAG}_cfg4_lag$lag.txt 2>&1
done
NPROJ=32 rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg4 --output-format csv -- python3 $R/scripts/prof_cfg4.py 30 40 4 > $OUT/${TAG}_cfg4_run.txt 2> $OUT/${TAG}_cfg4_run.err
python3 $R/scripts/trace_summary.py $OUT/prof_cfg4 --last 0.22 > $OUT/${TAG}_cfg4_trace_summary.txt
cat $OUT/${TAG}_cfg4_run.txt >> $OUT/${TAG}_cfg4_trace_summary.txt
NPROJ=32 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_cfg4_fetch --output-format csv -- python3 $R/scripts/prof_cfg4.py 30 12 2 > /dev/null 2> $OUT/${TAG}_cfg4_fetch.err
NPROJ=32 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_cfg4_write --output-format csv -- python3 $R/scripts/prof_cfg4.py 30 12 2 > /dev/null 2> $OUT/${TAG}_cfg4_write.err
python3 $R/scripts/pmc_summary.py $OUT/prof_cfg4_fetch $OUT/prof_cfg4_write $OUT/${TAG}_cfg4_pmc_fetch_write_per_kernel.json > $OUT/${TAG}_cfg4_pmc_summary.txt 2>&1
rm -rf $OUT/prof_cfg4 $OUT/prof_cfg4_fetch $OUT/prof_cfg4_write
tail -3 $OUT/${TAG}_cfg4_lag*.txt
cat $OUT/${TAG}_cfg4_trace_summary.txt
cat $OUT/${TAG}_cfg4_pmc_summary.txt
