#!/bin/bash
# Round 6, final tree: config 2's records -- graph-mode trace, k = 128 run with extras, the driver's command LAST (the quadrilateral
# kernel headers did not change since scripts/profile_r06.sh: its PMC table still covers them, bench.py checks the family hash)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 12 --warmup 10 --no-cpu-baseline --no-kdim --no-fortran-host"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_graph --output-format csv -- python3 $R/bench.py $ARGS > $O/r06_prof_graph.json 2> $O/r06_prof_graph.err
python3 $R/scripts/trace_summary.py $O/p_graph --last 0.5 > $O/r06_bench_graph_trace_summary.txt 2>&1
rm -rf $O/p_graph; head -16 $O/r06_bench_graph_trace_summary.txt
cd $R
python3 bench.py --case cfg3 --steps 4 --warmup 2 > $O/r06_bench_cfg3.json 2> $O/r06_bench_cfg3.err
python3 bench.py --extras --extras-out $O/r06_bench_extras.json > $O/r06_bench_k128.json 2> $O/r06_bench_k128.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
echo "driver command rc=$?"
python3 -c "
import json
for f in ('r06_bench_cfg3','r06_bench_k128','r06_bench'):
    r=json.load(open('$O/%s.json' % f)); print(f, 'value %.3f' % r['value'], 'ms/step %.2f' % r['ms_per_step'], 'ms/time step %.4f' % r['ms_per_time_step'], 'roofline', r['roofline'].get('frac'), r['roofline'].get('traffic'), 'e2e', (r.get('roofline_end_to_end') or {}).get('frac'), 'kdim', r.get('wall_time_kdim_s'), 'cpu', (r.get('cpu_baseline') or {}).get('value'), 'fortran', (r.get('fortran_host') or {}).get('matvecs_per_s'), 'budgets', (r.get('launch_budgets') or {}).get('per_time_step'), 'retries', r.get('map_retries'), 'ritz', (r.get('leading_ritz') or {}).get('re'), (r.get('leading_ritz') or {}).get('im'))"
