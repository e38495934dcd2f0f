#!/bin/bash
# Round-3 profiles of the bench command (run on the GPU box from the repository root):
#   kernel trace + stats, eager launches (and one attempt in hipGraph mode), PMC passes FETCH_SIZE / WRITE_SIZE (separate runs,
#   kernel trace only), summaries under gpurun_out/ -> copy into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out
ARGS="--steps 12 --warmup 2 --no-cpu-baseline --no-kdim"
export NSK_USE_GRAPH=0
rocprofv3 --kernel-trace --stats -d $OUT/prof_r03_eager --output-format csv -- python3 $R/bench.py $ARGS > $OUT/prof_r03_eager.json 2> $OUT/prof_r03_eager.err
python3 $R/scripts/trace_summary.py $OUT/prof_r03_eager --last 0.8 > $OUT/r03_bench_trace_summary.txt
cp $(ls $OUT/prof_r03_eager/*/*kernel_stats.csv | head -1) $OUT/r03_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_r03_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kdim > /dev/null 2> $OUT/prof_r03_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_r03_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kdim > /dev/null 2> $OUT/prof_r03_write.err
python3 $R/scripts/pmc_summary.py $OUT/prof_r03_fetch $OUT/prof_r03_write $OUT/r03_pmc_fetch_write_per_kernel.json > $OUT/r03_pmc_summary.txt 2>&1
python3 $R/scripts/kernel_table.py $OUT/prof_r03_eager $OUT/r03_pmc_fetch_write_per_kernel.json $OUT/prof_r03_eager.json > $OUT/r03_kernel_table.md 2> $OUT/r03_kernel_table.err
# the graph-replay timeline (what the un-profiled bench runs): rocprofv3 crashed inside hipGraph on this image in round 1
unset NSK_USE_GRAPH
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03_graph --output-format csv -- python3 $R/bench.py $ARGS > $OUT/prof_r03_graph.json 2> $OUT/prof_r03_graph.err
echo "graph-mode profile exit code $?" > $OUT/r03_graph_profile_status.txt
[ -d $OUT/prof_r03_graph ] && python3 $R/scripts/trace_summary.py $OUT/prof_r03_graph > $OUT/r03_bench_graph_trace_summary.txt 2>> $OUT/r03_graph_profile_status.txt
rm -rf $OUT/prof_r03_eager $OUT/prof_r03_fetch $OUT/prof_r03_write $OUT/prof_r03_graph     # raw traces are large; the summaries stay
ls -la $OUT | grep r03
