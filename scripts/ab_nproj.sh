#!/bin/bash
# size of the pressure projection space on the headline configuration (GPU box, repository root)
O=gpurun_out/r05; mkdir -p $O
for n in 32 24 16 8; do
  python3 bench.py --steps 60 --warmup 10 --nproj $n --no-cpu-baseline --no-fortran-host --no-kdim > $O/abn_$n.json 2> $O/abn_$n.err
  python3 -c "
import json; r=json.load(open('$O/abn_$n.json')); print('nproj $n', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'iters', r['helm_iters_per_step'], r['pres_iters_per_step'], r['launch_budgets']['per_time_step'], r['launch_budgets']['persistent_tail_maps'])"
done
