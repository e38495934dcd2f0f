#!/bin/bash
# GPU job 14: percentile budgets and no closing launch behind the tail (experimental library, options by environment)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
export NSK_LIB=$R/nekstab_amd/lib/libnekstab_hip_exp.so
run() { name=$1; shift
  env "$@" python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/ab14_$name.json 2> $O/ab14_$name.err
  python3 -c "
import json; r=json.load(open('$O/ab14_$name.json')); print('$name', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), 'budgets %.2f %.2f' % (r['launch_budgets']['per_time_step']['helm_launches_per_step'], r['launch_budgets']['per_time_step']['pres_iterations_per_step']), 'ritz %.12f %.12f' % (r['leading_ritz']['re'], r['leading_ritz']['im']))"
}
run base A=1
run close NSK_SKIP_CLOSE=1
run pct90 NSK_SB_PCT=90
run pct85 NSK_SB_PCT=85 NSK_SKIP_CLOSE=1
run pct90c NSK_SB_PCT=90 NSK_SKIP_CLOSE=1
run base A=1
run pct80c NSK_SB_PCT=80 NSK_SKIP_CLOSE=1
