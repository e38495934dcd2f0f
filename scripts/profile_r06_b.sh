#!/bin/bash
# Round-5 PMC passes of configs 4 and 3 on the final build (GPU box, repository root), merged into gpurun_out/r06/r06_pmc_traffic.json
# next to config 2's kernels (scripts/profile_r06.sh must have run: its r06_pmc_traffic.json is extended here).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/r06
mkdir -p $OUT
T=r06
K3="helm divgs schwarz schwarz_wg gs_dots8 gs_lag8 gs_dots24 gs_lag24 pres_rhs rhs convect_mfma"
REPS=6 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_c4f --output-format csv -- python3 $R/scripts/kernels3d_bench.py 30 $K3 > $OUT/${T}_cfg4_kernels_under_pmc.txt 2> $OUT/${T}_cfg4_fetch.err
REPS=6 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_c4w --output-format csv -- python3 $R/scripts/kernels3d_bench.py 30 $K3 > /dev/null 2> $OUT/${T}_cfg4_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_c4f $OUT/p_c4w $OUT/${T}_cfg4_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg4_pmc_summary.txt 2>&1
REPS=20 timeout 900 python3 $R/scripts/kernels3d_bench.py 30 > $OUT/${T}_cfg4_kernels.txt 2>&1
python3 $R/scripts/kernel_table_cfg4.py $OUT/${T}_cfg4_kernels.txt $OUT/${T}_cfg4_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg4_kernel_table.md 2> $OUT/${T}_cfg4_kernel_table.err
rm -rf $OUT/p_c4f $OUT/p_c4w
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_c3f --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${T}_cfg3_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_c3w --output-format csv -- python3 $R/bench.py --case cfg3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${T}_cfg3_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_c3f $OUT/p_c3w $OUT/${T}_cfg3_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg3_pmc_summary.txt 2>&1
rm -rf $OUT/p_c3f $OUT/p_c3w
true
cat $OUT/${T}_cfg4_kernel_table.md | head -30
# ---- config 5 (E = 99 452 hexahedra, lx1 = 10): kernel timings + PMC passes of the same launches
K5="helm divgs schwarz"
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="$K5" timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_c5f --output-format csv -- python3 $R/scripts/prof_cfg5.py 46 46 47 3 > $OUT/${T}_cfg5_kernels_under_pmc.txt 2> $OUT/${T}_cfg5_fetch.err
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="$K5" timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p_c5w --output-format csv -- python3 $R/scripts/prof_cfg5.py 46 46 47 3 > /dev/null 2> $OUT/${T}_cfg5_write.err
python3 $R/scripts/pmc_summary.py $OUT/p_c5f $OUT/p_c5w $OUT/${T}_cfg5_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg5_pmc_summary.txt 2>&1
rm -rf $OUT/p_c5f $OUT/p_c5w
SMOOTH=1 NPROJ=8 REPS=2 KERNELS="$K5" timeout 900 python3 $R/scripts/prof_cfg5.py 46 46 47 8 > $OUT/${T}_cfg5_steps.txt 2>&1
python3 $R/scripts/kernel_table_cfg5.py $OUT/${T}_cfg5_steps.txt $OUT/${T}_cfg5_pmc_fetch_write_per_kernel.json > $OUT/${T}_cfg5_kernel_table.md 2> $OUT/${T}_cfg5_kernel_table.err
python3 $R/scripts/pmc_traffic_merge_r06.py $OUT ${T} 2>&1 | tail -3
cat $OUT/${T}_cfg5_kernel_table.md
