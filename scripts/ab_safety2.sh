#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
run() {
  env "${@:2}" python3 bench.py --steps 118 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/abs_$1.json 2> $O/abs_$1.err
  python3 -c "
import json; r=json.load(open('$O/abs_$1.json')); print('$1', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], r['launch_budgets']['per_time_step'], 'tail maps', r['launch_budgets']['persistent_tail_maps'])"
}
run safety_2_1 NSK_TAIL=2 NSK_SB_HEAD_H=2 NSK_SB_HEAD_P=1
run safety_1_1 NSK_TAIL=2 NSK_SB_HEAD_H=1 NSK_SB_HEAD_P=1
run safety_2_0 NSK_TAIL=2 NSK_SB_HEAD_H=2 NSK_SB_HEAD_P=0
run safety_1_0 NSK_TAIL=2 NSK_SB_HEAD_H=1 NSK_SB_HEAD_P=0
run safety_3_1 NSK_TAIL=2 NSK_SB_HEAD_H=3 NSK_SB_HEAD_P=1
run safety_2_1b NSK_TAIL=2 NSK_SB_HEAD_H=2 NSK_SB_HEAD_P=1
