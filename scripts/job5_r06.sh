#!/bin/bash
# GPU job 5 of round 6: two-launch iteration, third cut: fp64 coarse image by default, solve start inside A_0, UC_EARLY variant
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_fuse2_gpu.py tests/test_persistent_gpu.py -x -q -s > $O/ab_fuse2_tests.txt 2>&1
tail -3 $O/ab_fuse2_tests.txt; grep -E "two vs three|rel L2|max \|H2|tail maps" $O/ab_fuse2_tests.txt | head -30
timeout 300 python3 scripts/pres_kernels_bench.py schwarz_uc3 schwarz_uc0 divgs_t pres_chain_merged pres_chain_fused gmres_update proj_apply > $O/ab_fuse2_kernels.txt 2>&1; cat $O/ab_fuse2_kernels.txt
NSK_LIB=$R/nekstab_amd/lib/libnekstab_hip_early.so timeout 300 python3 scripts/pres_kernels_bench.py schwarz_uc3 schwarz_uc0 pres_chain_fused 2>&1 | tail -3
run() { # name, env...
  name=$1; shift
  env "$@" python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/ab3_$name.json 2> $O/ab3_$name.err
  python3 -c "
import json; r=json.load(open('$O/ab3_$name.json')); print('$name', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), 'budgets %.2f %.2f' % (r['launch_budgets']['per_time_step']['helm_launches_per_step'], r['launch_budgets']['per_time_step']['pres_iterations_per_step']), r['launch_budgets']['persistent_tail_maps'], 'ritz %.12f %.12f %.3e' % (r['leading_ritz']['re'], r['leading_ritz']['im'], r['leading_ritz']['residual']))"
}
run three NSK_FUSE2=0
run two_nostart NSK_FUSE2_START=0
run two NSK_FUSE2_START=1
run two_early NSK_LIB=$R/nekstab_amd/lib/libnekstab_hip_early.so
run two_tail1 NSK_TAIL=1
run two NSK_FUSE2_START=1
run three NSK_FUSE2=0
