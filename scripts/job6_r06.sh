#!/bin/bash
# GPU job 6 of round 6: graph-mode kernel trace of the bench command on the two-launch build + the driver's command
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 12 --warmup 10 --no-cpu-baseline --no-kdim --no-fortran-host"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_graph --output-format csv -- python3 $R/bench.py $ARGS > $O/r06_prof_graph.json 2> $O/r06_prof_graph.err
python3 $R/scripts/trace_summary.py $O/p_graph --last 0.5 > $O/r06_bench_graph_trace_summary_mid.txt 2>&1
rm -rf $O/p_graph
cat $O/r06_bench_graph_trace_summary_mid.txt
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_mid.json 2> $O/r06_bench_mid.err
echo "driver command rc=$?"
python3 -c "
import json; r=json.load(open('$O/r06_bench_mid.json')); print('value %.3f' % r['value'], 'ms/step %.2f' % r['ms_per_step'], 'ms/time step %.4f' % r['ms_per_time_step'], 'roofline', r['roofline'].get('frac'), r['roofline'].get('avg_launch_us'), 'kdim', r.get('wall_time_kdim_s'), 'cpu', {k: r['cpu_baseline'].get(k) for k in ('value','cores_used','cores_visible','thread_calibration_ms_per_time_step')} if isinstance(r.get('cpu_baseline'), dict) else r.get('cpu_baseline'), 'fortran', (r.get('fortran_host') or {}).get('matvecs_per_s'), 'ritz', r['leading_ritz'])"
tail -5 $O/r06_bench_mid.err
