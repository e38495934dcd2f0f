#!/bin/bash
# Kernel trace of one config-4 Arnoldi step (315 time steps): per-kernel shares of the steady state (GPU box, repository root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/r05
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/p_c4t --output-format csv -- python3 $R/bench.py --case cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-kdim > $OUT/r05_cfg4_trace_bench.json 2> $OUT/r05_cfg4_trace.err
python3 $R/scripts/trace_summary.py $OUT/p_c4t --last 0.5 > $OUT/r05_cfg4_trace_summary.txt 2>&1
rm -rf $OUT/p_c4t
cat $OUT/r05_cfg4_trace_summary.txt | head -50
