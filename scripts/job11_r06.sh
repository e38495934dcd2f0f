#!/bin/bash
# GPU job 11 of round 6: the hexahedral bench lines with the shared-arrays-once roofline, config 3 with counters restricted to the time-step kernels
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
python3 bench.py --case cfg4 --steps 3 --warmup 1 > $O/r06_bench_cfg4.json 2> $O/r06_bench_cfg4.err
python3 bench.py --case cfg5 --steps 2 --warmup 1 > $O/r06_bench_cfg5.json 2> $O/r06_bench_cfg5.err
cd /tmp; export TMPDIR=/tmp
T=r06_cfg3
RX='k_helm|k_schwarz|k_divgs|k_rhs|k_pres_rhs|k_pres_update|k_vel_update|k_convect|k_proj|k_gmres|k_coarse'
timeout 900 rocprofv3 --kernel-trace --stats --kernel-include-regex "$RX" -d $O/p_c3t --output-format csv -- python3 $R/bench.py --case cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $O/${T}_prof_trace.json 2> $O/${T}_prof_trace.err
python3 $R/scripts/trace_summary.py $O/p_c3t --last 0.5 > $O/${T}_trace_summary.txt 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RX" -d $O/p_c3f --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/${T}_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RX" -d $O/p_c3w --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/${T}_write.err
python3 $R/scripts/pmc_summary.py $O/p_c3f $O/p_c3w $O/${T}_pmc_fetch_write_per_kernel.json > $O/${T}_pmc_summary.txt 2>&1
python3 $R/scripts/kernel_table.py $O/p_c3t $O/${T}_pmc_fetch_write_per_kernel.json $O/${T}_prof_trace.json > $O/${T}_kernel_table.md 2> $O/${T}_kernel_table.err
rm -rf $O/p_c3t $O/p_c3f $O/p_c3w
python3 $R/scripts/pmc_traffic_merge_r06.py $O r06 2>&1 | tail -2
tail -3 $O/${T}_fetch.err; cat $O/${T}_pmc_summary.txt | tail -12; head -16 $O/${T}_kernel_table.md
cd $R; python3 bench.py --case cfg3 --steps 4 --warmup 2 > $O/r06_bench_cfg3.json 2> $O/r06_bench_cfg3.err
python3 -c "
import json
for f in ('cfg3','cfg4','cfg5'):
    r=json.load(open('$O/r06_bench_%s.json' % f)); print(f, 'value %.4f' % r['value'], 'ms/time step %.3f' % r['ms_per_time_step'], r['roofline'].get('frac'), r['roofline'].get('traffic'))"
