#!/bin/bash
# GPU job 28: borrowed captured steps while a budget pair is young (gc_slack) -- the driver's timed window (Arnoldi steps 5-24) and a
# (the gc_slack option of this job ran on an experimental build: NOT in the tree -- DESIGN.md section 7)
# 60-step window, alternating on one box; iteration counts and Ritz values must not move
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
run() { name=$1; st=$2; wu=$3; shift 3
  env "$@" python3 bench.py --steps $st --warmup $wu --no-cpu-baseline --no-fortran-host --no-kdim > $O/ab28_$name.json 2> $O/ab28_$name.err
  python3 -c "
import json; r=json.load(open('$O/ab28_$name.json')); print('$name', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'captures', r['graph_recaptures'], '%.1f ms' % (1e3*r['graph_recapture_s']), 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), 'budgets %.2f %.2f' % (r['launch_budgets']['per_time_step']['helm_launches_per_step'], r['launch_budgets']['per_time_step']['pres_iterations_per_step']))"
}
run exact_a 20 5 NSK_GC_SLACK=0
run slack_a 20 5 NSK_GC_SLACK=1
run exact_b 20 5 NSK_GC_SLACK=0
run slack_b 20 5 NSK_GC_SLACK=1
run slack16 20 5 NSK_GC_SLACK=1 NSK_GC_MIN_WANT=16
run slack256 20 5 NSK_GC_SLACK=1 NSK_GC_MIN_WANT=256
run exact60 60 10 NSK_GC_SLACK=0
run slack60 60 10 NSK_GC_SLACK=1
