#!/bin/bash
# Round 6, the very last tree: the whole GPU suite, then the driver's command
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu --durations=12 > $O/r06_gputest.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $O/r06_gputest.txt | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
echo "driver command rc=$?"
python3 -c "
import json
r=json.load(open('$O/r06_bench.json')); print('value %.3f' % r['value'], 'ms/step %.2f' % r['ms_per_step'], 'roofline', r['roofline'].get('frac'), r['roofline'].get('traffic'), r['roofline'].get('traffic_source','')[:90], 'kdim', r.get('wall_time_kdim_s'), 'cpu', (r.get('cpu_baseline') or {}).get('value'), 'fortran', (r.get('fortran_host') or {}).get('matvecs_per_s'))"
