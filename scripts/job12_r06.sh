#!/bin/bash
# GPU job 12 of round 6 (final tree): the whole GPU suite, config 3's bench line (traffic from this build's counters), the extras run (k = 128 + A/B legs), the driver's command LAST
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=25 > $O/r06_gputest.txt 2>&1; tail -3 $O/r06_gputest.txt
python3 bench.py --case cfg3 --steps 4 --warmup 2 > $O/r06_bench_cfg3.json 2> $O/r06_bench_cfg3.err
python3 bench.py --extras --extras-out $O/r06_bench_extras.json > $O/r06_bench_k128.json 2> $O/r06_bench_k128.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
echo "driver command rc=$?"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06/r06_bench*.json')):
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    if 'value' not in r: continue
    print(f, 'value %.3f' % r['value'], 'ms/step %.1f' % r['ms_per_step'], 'ms/time step', r.get('ms_per_time_step'), 'roofline', r['roofline'].get('frac'), 'traffic', r['roofline'].get('traffic'),
          'e2e', r.get('roofline_end_to_end', {}).get('frac'), 'kdim', r.get('wall_time_kdim_s'), 'cpu', (r.get('cpu_baseline') or {}).get('value'), 'fortran', (r.get('fortran_host') or {}).get('matvecs_per_s'))
PY
