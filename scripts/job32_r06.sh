#!/bin/bash
# GPU job 32: the sharded hexahedral step with k_helm_p<10> (virtual ranks) and the multi-process tests
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_sharded_gpu.py tests/test_multiprocess_gpu.py tests/test_sharded_r3_gpu.py -q -x > $O/job32.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" $O/job32.txt | tail -3
