#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
for v in 0 8 0 8; do
  NSK_LOADS_FIRST=$v python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/lf8_$v.json 2> $O/lf8_$v.err
  python3 -c "
import json; r=json.load(open('$O/lf8_$v.json')); print('lf=$v', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'helm %.2f us' % r['roofline']['avg_launch_us'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']))"
done
