#!/bin/bash
# config 3 (E = 7984 quadrilaterals, lx1 = 12): captured step graphs with launch budgets against eager steps with host-read
# convergence flags (option "hostcheck"): Arnoldi steps per second, iterations, redone maps
R=${GRAFT_REPO_ROOT:-$PWD}
for hc in 0 1; do
  NSK_HOSTCHECK=$hc python3 $R/bench.py --case cfg3 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('hostcheck $hc', {k:r[k] for k in ('value','ms_per_step','helm_iters_per_step','pres_iters_per_step','map_retries','graph_recaptures')})"
done
