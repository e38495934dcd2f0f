#!/bin/bash
# GPU job 13: k_helm specialised on it >= 2 (experimental library) against the production library
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
L=$R/nekstab_amd/lib/libnekstab_hip_helm2.so
for lib in prod exp prod exp; do
  if [ $lib = exp ]; then export NSK_LIB=$L; else unset NSK_LIB; fi
  python3 scripts/pres_kernels_bench.py helm 2>&1 | grep -E "^helm" | sed "s/^/$lib /"
  python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/ab_helm2_$lib.json 2> $O/ab_helm2_$lib.err
  python3 -c "
import json; r=json.load(open('$O/ab_helm2_$lib.json')); print('$lib', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), 'helm us', r['roofline']['avg_launch_us'], 'ritz %.12f %.12f' % (r['leading_ritz']['re'], r['leading_ritz']['im']))"
done
NSK_LIB=$L python3 -m pytest tests/test_persistent_gpu.py -q -x -k "tails" 2>&1 | tail -3
