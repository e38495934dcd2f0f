#!/bin/bash
# Round 6, second half (the lx1 = 10 kernels): counters and tables of configs 4 and 5 on THIS build (the hexahedral kernel headers
# changed, so round 6's first PMC table no longer covers them), merged into r06_pmc_traffic.json next to config 2's and config 3's
# entries (quadrilateral headers unchanged), then the bench lines of configs 4 and 5.   GPU box, repository root.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06; mkdir -p $O
cp $R/profiles/r06_pmc_traffic.json $O/r06_pmc_traffic.json
cd /tmp; export TMPDIR=/tmp
T=r06
# ---- config 4 (E = 50 100 hexahedra, lx1 = 8): unchanged kernels, this build's counters
K3="helm divgs schwarz schwarz_wg gs_dots8 gs_lag8 gs_dots24 gs_lag24 pres_rhs rhs convect_mfma"
REPS=6 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/p_c4f --output-format csv -- python3 $R/scripts/kernels3d_bench.py 30 $K3 > $O/${T}_cfg4_kernels_under_pmc.txt 2> $O/${T}_cfg4_fetch.err
REPS=6 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/p_c4w --output-format csv -- python3 $R/scripts/kernels3d_bench.py 30 $K3 > /dev/null 2> $O/${T}_cfg4_write.err
python3 $R/scripts/pmc_summary.py $O/p_c4f $O/p_c4w $O/${T}_cfg4_pmc_fetch_write_per_kernel.json > $O/${T}_cfg4_pmc_summary.txt 2>&1
REPS=20 timeout 900 python3 $R/scripts/kernels3d_bench.py 30 > $O/${T}_cfg4_kernels.txt 2>&1
python3 $R/scripts/kernel_table_cfg4.py $O/${T}_cfg4_kernels.txt $O/${T}_cfg4_pmc_fetch_write_per_kernel.json > $O/${T}_cfg4_kernel_table.md 2> $O/${T}_cfg4_kernel_table.err
rm -rf $O/p_c4f $O/p_c4w
head -16 $O/${T}_cfg4_kernel_table.md
# ---- config 5 (E = 99 452 hexahedra, lx1 = 10): the new forms and the old ones, timings + PMC passes of the same launches
K5="helm helm_wg divgs divgs_wg schwarz schwarz_p schwarz_wg convect_mfma convect_mfma_nl"
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="$K5" timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/p_c5f --output-format csv -- python3 $R/scripts/prof_cfg5.py 46 46 47 3 > $O/${T}_cfg5_kernels_under_pmc.txt 2> $O/${T}_cfg5_fetch.err
SMOOTH=1 NPROJ=8 REPS=1 KERNELS="$K5" timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/p_c5w --output-format csv -- python3 $R/scripts/prof_cfg5.py 46 46 47 3 > /dev/null 2> $O/${T}_cfg5_write.err
python3 $R/scripts/pmc_summary.py $O/p_c5f $O/p_c5w $O/${T}_cfg5_pmc_fetch_write_per_kernel.json > $O/${T}_cfg5_pmc_summary.txt 2>&1
rm -rf $O/p_c5f $O/p_c5w
SMOOTH=1 NPROJ=8 REPS=2 KERNELS="$K5 convect convect_nl" timeout 900 python3 $R/scripts/prof_cfg5.py 46 46 47 8 > $O/${T}_cfg5_steps.txt 2>&1
python3 $R/scripts/kernel_table_cfg5.py $O/${T}_cfg5_steps.txt $O/${T}_cfg5_pmc_fetch_write_per_kernel.json > $O/${T}_cfg5_kernel_table.md 2> $O/${T}_cfg5_kernel_table.err
python3 $R/scripts/pmc_traffic_merge_r06.py $O ${T} 2>&1 | tail -3
cat $O/${T}_cfg5_kernel_table.md
# ---- the bench lines (traffic from the table just written: it travels with this call only, so they run here)
cd $R
cp $O/r06_pmc_traffic.json $R/profiles/r06_pmc_traffic.json
python3 bench.py --case cfg4 --steps 3 --warmup 1 > $O/r06_bench_cfg4.json 2> $O/r06_bench_cfg4.err
python3 bench.py --case cfg5 --steps 2 --warmup 1 > $O/r06_bench_cfg5.json 2> $O/r06_bench_cfg5.err
python3 -c "
import json
for f in ('cfg4','cfg5'):
    r=json.load(open('$O/r06_bench_%s.json' % f)); print(f, 'value %.4f' % r['value'], 'ms/time step %.3f' % r['ms_per_time_step'], r['roofline'].get('frac'), r['roofline'].get('traffic'), r.get('kernel_us'))"
