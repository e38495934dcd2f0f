#!/bin/bash
# GPU job 27: the whole GPU suite on the lx1 = 10 kernels
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu --durations=12 > $O/r06_gputest_b.txt 2>&1
tail -25 $O/r06_gputest_b.txt
