#!/bin/bash
# Round-4 record runs (GPU box, repository root): the rocprofv3 recipe, then the un-profiled bench lines of configs 2, 3, 4 and
# the full-size eigen run of config 3.  Everything lands in gpurun_out/ (copy into profiles/).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash scripts/profile_r04.sh > gpurun_out/r04_profile_log.txt 2>&1
python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python3 bench.py --case cfg3 --steps 4 --warmup 2 > gpurun_out/r04_bench_cfg3.json 2> gpurun_out/r04_bench_cfg3.err
python3 bench.py --case cfg4 --steps 3 --warmup 1 > gpurun_out/r04_bench_cfg4.json 2> gpurun_out/r04_bench_cfg4.err
python3 scripts/run_cfg3_eigen.py 64 2 > gpurun_out/r04_cfg3_eigen.txt 2>&1
SMOOTH=1 NPROJ=8 REPS=2 python3 scripts/prof_cfg5.py 46 46 47 10 > gpurun_out/r04_cfg5_steps.txt 2>&1
tail -2 gpurun_out/r04_cfg3_eigen.txt gpurun_out/r04_cfg5_steps.txt
python3 -c "
import json
for f in ('r04_bench','r04_bench_cfg3','r04_bench_cfg4'):
    r=json.load(open('gpurun_out/%s.json'%f)); print(f, r['value'], r['ms_per_step'], r.get('ms_per_time_step'), r['roofline']['frac'], r['roofline'].get('traffic'), r.get('roofline_end_to_end',{}).get('frac'), r.get('wall_time_kdim_s'))
"
