#!/bin/bash
# Round-4 profiles (run on the GPU box from the repository root; summaries land in gpurun_out/, copy them into profiles/):
#   A. config 2 (the driver's bench command): eager kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel
#      trace only), per-kernel table, graph-mode trace
#   B. config 4 at full size: steady-state kernel trace of scripts/prof_cfg4.py (last of four 40-step maps), FETCH_SIZE /
#      WRITE_SIZE passes of scripts/kernels3d_bench.py (every hot hexahedral kernel launched back to back at a KNOWN basis index,
#      so the counter value per launch is exact), per-kernel table
#   C. config 3: FETCH_SIZE / WRITE_SIZE of k_helm<12>
# and profiles-style r04_pmc_traffic.json (bytes per launch keyed by kernel, stamped with the library's source hash).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out
T=r04
ARGS="--steps 12 --warmup 2 --no-cpu-baseline --no-kdim"
# ---- A
NSK_USE_GRAPH=0 rocprofv3 --kernel-trace --stats -d $OUT/p_eager --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${T}_prof_eager.json 2> $OUT/${T}_prof_eager.err
python3 $R/scripts/trace_summary.py $OUT/p_eager --last 0.8 > $OUT/${T}_bench_trace_summary.txt
cp $(ls $OUT/p_eager/*/*kernel_stats.csv | head -1) $OUT/${T}_bench_kernel_stats.csv
NSK_USE_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kdim > /dev/null 2> $OUT/${T}_prof_fetch.err
NSK_USE_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kdim > /dev/null 2> $OUT/${T}_prof_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_fetch $OUT/p_write $OUT/${T}_pmc_fetch_write_per_kernel.json > $OUT/${T}_pmc_summary.txt 2>&1
python3 $R/scripts/kernel_table.py $OUT/p_eager $OUT/${T}_pmc_fetch_write_per_kernel.json $OUT/${T}_prof_eager.json > $OUT/${T}_kernel_table.md 2> $OUT/${T}_kernel_table.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_graph --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${T}_prof_graph.json 2> $OUT/${T}_prof_graph.err
[ -d $OUT/p_graph ] && python3 $R/scripts/trace_summary.py $OUT/p_graph > $OUT/${T}_bench_graph_trace_summary.txt 2>&1
rm -rf $OUT/p_eager $OUT/p_fetch $OUT/p_write $OUT/p_graph
# ---- B
NPROJ=32 rocprofv3 --kernel-trace --stats -d $OUT/p_cfg4 --output-format csv -- python3 $R/scripts/prof_cfg4.py 30 40 4 > $OUT/${T}_cfg4_run.txt 2> $OUT/${T}_cfg4_run.err
python3 $R/scripts/trace_summary.py $OUT/p_cfg4 --last 0.22 > $OUT/${T}_cfg4_trace_summary.txt
cat $OUT/${T}_cfg4_run.txt >> $OUT/${T}_cfg4_trace_summary.txt
K3="helm divgs schwarz schwarz_wg gs_dots8 gs_lag8 gs_dots24 gs_lag24 pres_rhs rhs convect_mfma"
REPS=6 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_c4f --output-format csv -- python3 $R/scripts/kernels3d_bench.py 30 $K3 > $OUT/${T}_cfg4_kernels_under_pmc.txt 2> $OUT/${T}_cfg4_fetch.err
REPS=6 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_c4w --output-format csv -- python3 $R/scripts/kernels3d_bench.py 30 $K3 > /dev/null 2> $OUT/${T}_cfg4_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_c4f $OUT/p_c4w $OUT/${T}_cfg4_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg4_pmc_summary.txt 2>&1
REPS=20 python3 $R/scripts/kernels3d_bench.py 30 > $OUT/${T}_cfg4_kernels.txt 2>&1
python3 $R/scripts/kernel_table_cfg4.py $OUT/${T}_cfg4_kernels.txt $OUT/${T}_cfg4_pmc_fetch_write_per_kernel.json $OUT/p_cfg4 > $OUT/${T}_cfg4_kernel_table.md 2> $OUT/${T}_cfg4_kernel_table.err
rm -rf $OUT/p_cfg4 $OUT/p_c4f $OUT/p_c4w
# ---- C
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_c3f --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${T}_cfg3_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_c3w --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${T}_cfg3_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_c3f $OUT/p_c3w $OUT/${T}_cfg3_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg3_pmc_summary.txt 2>&1
rm -rf $OUT/p_c3f $OUT/p_c3w
python3 $R/scripts/pmc_traffic_merge.py $OUT ${T}
ls -la $OUT | grep ${T}_
