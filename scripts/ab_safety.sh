#!/bin/bash
# per-step budgets alone against budgets + persistent tail as a safety net, k = 128 runs (GPU box, repository root)
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  env "${@:2}" python3 bench.py --steps 118 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/abs_$1.json 2> $O/abs_$1.err
  python3 -c "
import json; r=json.load(open('$O/abs_$1.json')); print('$1', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], r['launch_budgets']['per_time_step'], 'tail maps', r['launch_budgets']['persistent_tail_maps'])"
}
run budgets_3_2 NSK_TAIL=0
run budgets_3_3 NSK_TAIL=0 NSK_SB_HEAD_P=3
run safety_3_2 NSK_TAIL=2
run safety_2_1 NSK_TAIL=2 NSK_SB_HEAD_H=2 NSK_SB_HEAD_P=1
run safety_2_2 NSK_TAIL=2 NSK_SB_HEAD_H=2 NSK_SB_HEAD_P=2
run tails_median NSK_TAIL=-1
