#!/bin/bash
# A/B of the persistent tails on the headline configuration (GPU box, repository root)
O=gpurun_out/r05; mkdir -p $O
for v in -1 0; do
  NSK_TAIL=$v python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/ab_tail$v.json 2> $O/ab_tail$v.err
  python3 -c "
import json; r=json.load(open('$O/ab_tail$v.json')); print('tail=$v', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'recaptures', r['graph_recaptures'], r['launch_budgets']['per_time_step'], r['launch_budgets']['persistent_tail_maps'], 'iters', r['helm_iters_per_step'], r['pres_iters_per_step'], r['leading_ritz']['re'], r['leading_ritz']['residual'])"
done
