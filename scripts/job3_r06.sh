#!/bin/bash
# GPU job 3 of round 6: stamps of the two-launch iteration, the tight-tolerance fuse2 test, the Hessenberg fixture
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 300 python3 scripts/stamps_fuse2.py 3 > $O/stamps_fuse2_j3.txt 2>&1; cat $O/stamps_fuse2_j3.txt
timeout 300 python3 scripts/stamps_fuse2.py 1 > $O/stamps_fuse2_j1.txt 2>&1; cat $O/stamps_fuse2_j1.txt
timeout 300 python3 -m pytest tests/test_fuse2_gpu.py -x -q -s -k "1e-08 or oracle or arnoldi" > $O/fuse2_tight.txt 2>&1; grep -E "pressure iterations per step|two vs three|rel L2|max \|H2|passed|failed|Error" $O/fuse2_tight.txt
timeout 300 python3 tests/golden/make_hessenberg_fixture.py > $O/hess_fixture.txt 2>&1; tail -2 $O/hess_fixture.txt
