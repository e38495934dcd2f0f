#!/bin/bash
# kernel trace of the bench command with the persistent tails on (GPU box, repository root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/r05
mkdir -p $OUT
NSK_TAIL=${NSK_TAIL:--1} timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p_tail --output-format csv -- python3 $R/bench.py --steps 12 --warmup 10 --no-cpu-baseline --no-kdim --no-fortran-host > $OUT/r05_prof_tail.json 2> $OUT/r05_prof_tail.err
python3 $R/scripts/trace_summary.py $OUT/p_tail --last 0.5 > $OUT/r05_tail_trace_summary.txt 2>&1
python3 - <<PY
import csv, glob, numpy as np
f = glob.glob('$OUT/p_tail/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
t1 = max(int(r['End_Timestamp']) for r in rows); t0 = min(int(r['Start_Timestamp']) for r in rows)
rows = [r for r in rows if int(r['Start_Timestamp']) > t1 - 0.5 * (t1 - t0)]
for name in ('k_helm_tail', 'k_pres_tail'):
    v = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if name in r['Kernel_Name']])
    if len(v): print(name, 'calls', len(v), 'mean %.2f' % v.mean(), 'percentiles 10/25/50/75/90/99:', np.percentile(v, [10, 25, 50, 75, 90, 99]).round(2))
PY
rm -rf $OUT/p_tail
cat $OUT/r05_tail_trace_summary.txt | head -24
