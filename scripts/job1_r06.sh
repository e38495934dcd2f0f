#!/bin/bash
# first GPU job of round 6: CPU-port thread scaling on the box's host cores, the wake-row artefact
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r06
mkdir -p $O
nproc > $O/r06_nproc.txt; lscpu >> $O/r06_nproc.txt
timeout 1500 python3 scripts/cpu_scaling.py --threads 4,8,16,32,64,128 > $O/r06_cpu_scaling_v0.txt 2>&1
tail -8 $O/r06_cpu_scaling_v0.txt
timeout 900 python3 scripts/wake_rows.py > $O/r06_wake_rows.log 2>&1
tail -12 $O/r06_wake_rows.log
