#!/bin/bash
# the bench command without the CPU legs, twice (GPU box, repository root): value, ms per time step, iteration counts
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
  python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/quick_$rep.json 2> $O/quick_$rep.err
  python3 -c "
import json; r=json.load(open('$O/quick_$rep.json')); print('run $rep', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), r['launch_budgets']['per_time_step'], r['launch_budgets']['persistent_tail_maps'], 'roofline %.3f %.2f us' % (r['roofline']['frac'], r['roofline']['avg_launch_us']), 'ritz %.10f %.3e' % (r['leading_ritz']['re'], r['leading_ritz']['residual']))"
done
