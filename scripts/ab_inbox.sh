#!/bin/bash
# dssum inboxes on / off x "loads first" masks (GPU box, repository root)
O=gpurun_out/r05; mkdir -p $O
run() {
  env "${@:2}" python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/abi_$1.json 2> $O/abi_$1.err
  python3 -c "
import json; r=json.load(open('$O/abi_$1.json')); print('$1', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), 'helm %.2f us frac %.3f' % (r['roofline']['avg_launch_us'], r['roofline']['frac']), 'ritz %.12f' % r['leading_ritz']['re'])"
}
run noinbox_lf0 NSK_NO_INBOX=1 NSK_LOADS_FIRST=0
run inbox_lf0 NSK_LOADS_FIRST=0
run inbox_lf2 NSK_LOADS_FIRST=2
run inbox_lf4 NSK_LOADS_FIRST=4
run inbox_lf6 NSK_LOADS_FIRST=6
run inbox_lf6_tail1 NSK_LOADS_FIRST=6 NSK_TAIL=1
run inbox_lf7_tail1 NSK_LOADS_FIRST=7 NSK_TAIL=1
run inbox_lf0b NSK_LOADS_FIRST=0
