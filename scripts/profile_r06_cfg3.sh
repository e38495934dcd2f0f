#!/bin/bash
# Round-6 profiles of config 3 (E = 7984 quadrilaterals, lx1 = 12): kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes
# (separate runs with --kernel-trace only), per-kernel table.  GPU box, repository root; results in gpurun_out/r06/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/r06
mkdir -p $OUT
T=r06_cfg3
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/p_c3t --output-format csv -- python3 $R/bench.py --case cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${T}_prof_trace.json 2> $OUT/${T}_prof_trace.err
python3 $R/scripts/trace_summary.py $OUT/p_c3t --last 0.5 > $OUT/${T}_trace_summary.txt 2>&1
cp $(ls $OUT/p_c3t/*/*kernel_stats.csv | head -1) $OUT/${T}_kernel_stats.csv
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_c3f --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${T}_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_c3w --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${T}_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_c3f $OUT/p_c3w $OUT/${T}_pmc_fetch_write_per_kernel.json > $OUT/${T}_pmc_summary.txt 2>&1
python3 $R/scripts/kernel_table.py $OUT/p_c3t $OUT/${T}_pmc_fetch_write_per_kernel.json $OUT/${T}_prof_trace.json > $OUT/${T}_kernel_table.md 2> $OUT/${T}_kernel_table.err
rm -rf $OUT/p_c3t $OUT/p_c3f $OUT/p_c3w
ls -la $OUT | grep ${T}_
