#!/bin/bash
# GPU job 24: k_schwarz_q<10> (four wavefronts per element) -- parity with the other forms, timing at 24^3 elements
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python3 -m pytest tests/test_3d_gpu.py -q -x -k "forms_agree" 2>&1 | tail -4
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="schwarz_wg schwarz_p schwarz_q schwarz_q divgs" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | tail -8
