#!/bin/bash
# GPU job 26: k_helm_p<10> with the next element's per-node constants loaded one element ahead (all ten: 44 B of scratch; four of
# (hip_exp / hip_exp2: experimental builds with -DNSK_HP_GEO=1 / 2, NOT in the tree -- DESIGN.md section 7)
# them: none) against the committed form, 24^3 elements; bit identity of the variants by the tails of a 3-step map
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for v in hip hip_exp hip_exp2 hip hip_exp hip_exp2; do
  echo "== $v"; NSK_LIB=$R/nekstab_amd/lib/libnekstab_$v.so SMOOTH=1 NPROJ=8 REPS=1 KERNELS="helm helm" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | grep -E "helm|per step"
done
