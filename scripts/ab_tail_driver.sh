#!/bin/bash
# the driver's command (5 + 20 Arnoldi steps) with the persistent tails on / off, three times each (GPU box, repository root)
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do for v in -1 0; do
  NSK_TAIL=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fortran-host --no-kdim > $O/abd_tail$v.json 2> $O/abd_tail$v.err
  python3 -c "
import json; r=json.load(open('$O/abd_tail$v.json')); print('tail=$v', 'value %.3f' % r['value'], 'retries', r['map_retries'], r['launch_budgets']['per_time_step'], r['launch_budgets']['persistent_tail_maps'])"
done; done
