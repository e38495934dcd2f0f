#!/bin/bash
# GPU job 23: where k_helm_p<10>'s time goes -- the kernel without its matrix-core passes, without the LDS-DMA prefetch, without both
# (timing experiments with wrong values: experimental libraries), 24^3 elements
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for v in hip hip_exp hip_exp2 hip_exp3; do
  echo "== $v"; NSK_LIB=$R/nekstab_amd/lib/libnekstab_$v.so SMOOTH=1 NPROJ=8 REPS=1 KERNELS="helm helm helm_wg" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | grep -E "helm"
done
