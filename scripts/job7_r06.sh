#!/bin/bash
# GPU job 7 of round 6: the whole GPU suite on the two-launch build
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=25 > $O/r06_gputest_mid.txt 2>&1
tail -45 $O/r06_gputest_mid.txt
