#!/bin/bash
# variants of scripts/repro_r04_nan.py (GPU box, repository root); outputs under gpurun_out/r05/
mkdir -p gpurun_out/r05
for v in "$@"; do
  n=$(echo $v | tr " ,=" "___")
  python3 scripts/repro_r04_nan.py $v > gpurun_out/r05/rep2_$n.txt 2>&1
done
grep -H -v amdgpu.ids gpurun_out/r05/rep2_*.txt | tail -80
