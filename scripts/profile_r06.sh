#!/bin/bash
# Round-5 profiles of config 2 (GPU box, repository root; summaries land in gpurun_out/r06/, copy them into profiles/):
#   graph-mode kernel trace of the bench command (what the driver times), eager trace + stats, FETCH_SIZE / WRITE_SIZE passes
#   (separate runs, kernel trace only), per-kernel table and r06_pmc_traffic.json (bytes per launch, stamped with the source hash)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/r06
mkdir -p $OUT
T=r06
ARGS="--steps 12 --warmup 10 --no-cpu-baseline --no-kdim --no-fortran-host"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p_graph --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${T}_prof_graph.json 2> $OUT/${T}_prof_graph.err
[ -d $OUT/p_graph ] && python3 $R/scripts/trace_summary.py $OUT/p_graph --last 0.5 > $OUT/${T}_bench_graph_trace_summary.txt 2>&1
NSK_USE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p_eager --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${T}_prof_eager.json 2> $OUT/${T}_prof_eager.err
python3 $R/scripts/trace_summary.py $OUT/p_eager --last 0.5 > $OUT/${T}_bench_trace_summary.txt
cp $(ls $OUT/p_eager/*/*kernel_stats.csv | head -1) $OUT/${T}_bench_kernel_stats.csv
NSK_USE_GRAPH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kdim --no-fortran-host > /dev/null 2> $OUT/${T}_prof_fetch.err
NSK_USE_GRAPH=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kdim --no-fortran-host > /dev/null 2> $OUT/${T}_prof_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_fetch $OUT/p_write $OUT/${T}_pmc_fetch_write_per_kernel.json > $OUT/${T}_pmc_summary.txt 2>&1
python3 $R/scripts/kernel_table.py $OUT/p_eager $OUT/${T}_pmc_fetch_write_per_kernel.json $OUT/${T}_prof_eager.json > $OUT/${T}_kernel_table.md 2> $OUT/${T}_kernel_table.err
rm -rf $OUT/p_eager $OUT/p_fetch $OUT/p_write $OUT/p_graph
true
ls -la $OUT | grep ${T}_
cat $OUT/${T}_bench_graph_trace_summary.txt
