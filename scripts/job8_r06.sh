#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
python3 -m pytest tests/test_persistent_gpu.py -q -x -s -k tails 2>&1 | grep -E "reference run|tail maps|passed|failed|assert" | head -20
timeout 300 python3 scripts/pres_kernels_bench.py proj_update proj_apply proj_apply_e pres_rhs pres_update vel_update_proj rhs convect helm schwarz_uc3 divgs_t pres_chain_fused 2>&1 | grep -v amdgpu
