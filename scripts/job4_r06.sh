#!/bin/bash
# GPU job 4 of round 6: second cut of the two-launch iteration (compile-time bounds, prefetch stages, fp32 coarse image): tests, stamps, kernel timings, bench A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_fuse2_gpu.py tests/test_persistent_gpu.py -x -q -s > $O/ab_fuse2_tests.txt 2>&1
tail -3 $O/ab_fuse2_tests.txt; grep -E "two vs three|rel L2|max \|H2|tail maps" $O/ab_fuse2_tests.txt | head -30
timeout 300 python3 scripts/stamps_fuse2.py 3 > $O/stamps_fuse2_j3.txt 2>&1; cat $O/stamps_fuse2_j3.txt
timeout 300 python3 scripts/pres_kernels_bench.py update_coarse3 schwarz divgs2 schwarz_uc3 schwarz_uc0 divgs_t pres_chain_merged pres_chain_fused gmres_update > $O/ab_fuse2_kernels.txt 2>&1
cat $O/ab_fuse2_kernels.txt
NSK_TC32=0 timeout 300 python3 scripts/pres_kernels_bench.py divgs_t pres_chain_fused 2>&1 | tail -2
for f2 in 0 1 1; do
  NSK_FUSE2=$f2 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fortran-host --no-kdim > $O/ab_fuse2_$f2.json 2> $O/ab_fuse2_$f2.err
  python3 -c "
import json; r=json.load(open('$O/ab_fuse2_$f2.json')); print('fuse2 $f2', 'value %.3f' % r['value'], 'ms/time step %.4f' % r['ms_per_time_step'], 'retries', r['map_retries'], 'iters %.4f %.4f' % (r['helm_iters_per_step'], r['pres_iters_per_step']), r['launch_budgets']['per_time_step'], r['launch_budgets']['persistent_tail_maps'], 'ritz %.12f %.12f %.3e' % (r['leading_ritz']['re'], r['leading_ritz']['im'], r['leading_ritz']['residual']))"
done
