#!/bin/bash
# GPU job 22: k_helm_p<10> with non-temporal stores and LDS-DMA loads (experimental library) against the committed form, 24^3 elements
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
for v in base nt base nt; do
  L=$R/nekstab_amd/lib/libnekstab_hip.so; [ $v = nt ] && L=$R/nekstab_amd/lib/libnekstab_hip_exp.so
  echo "== $v"; NSK_LIB=$L SMOOTH=1 NPROJ=8 REPS=1 KERNELS="helm helm" timeout 600 python3 scripts/prof_cfg5.py 24 24 24 3 2>&1 | grep -E "ms per step|helm"
done
