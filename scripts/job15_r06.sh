#!/bin/bash
# GPU job 15 of round 6 (final tree, percentile budgets + no closing launch behind the tail): tests of the budget / tail machinery, graph-mode trace, k = 128 run with extras, the driver's command LAST
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_persistent_gpu.py tests/test_fuse2_gpu.py tests/test_bench_driver_gpu.py tests/test_errors_gpu.py tests/test_krylov_gpu.py -q -x 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 12 --warmup 10 --no-cpu-baseline --no-kdim --no-fortran-host"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_graph --output-format csv -- python3 $R/bench.py $ARGS > $O/r06_prof_graph.json 2> $O/r06_prof_graph.err
python3 $R/scripts/trace_summary.py $O/p_graph --last 0.5 > $O/r06_bench_graph_trace_summary.txt 2>&1
rm -rf $O/p_graph; head -16 $O/r06_bench_graph_trace_summary.txt
cd $R
python3 bench.py --extras --extras-out $O/r06_bench_extras.json > $O/r06_bench_k128.json 2> $O/r06_bench_k128.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
echo "driver command rc=$?"
python3 -c "
import json
for f in ('r06_bench_k128','r06_bench'):
    r=json.load(open('$O/%s.json' % f)); print(f, 'value %.3f' % r['value'], 'ms/step %.2f' % r['ms_per_step'], 'ms/time step %.4f' % r['ms_per_time_step'], 'roofline', r['roofline'].get('frac'), r['roofline'].get('traffic'), 'e2e', r['roofline_end_to_end']['frac'], 'kdim', r.get('wall_time_kdim_s'), 'cpu', (r.get('cpu_baseline') or {}).get('value'), 'fortran', (r.get('fortran_host') or {}).get('matvecs_per_s'), 'budgets', r['launch_budgets']['per_time_step'], 'retries', r['map_retries'], 'ritz', r['leading_ritz']['re'], r['leading_ritz']['im'])"
