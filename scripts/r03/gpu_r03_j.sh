#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/r03j_touch_first.log
: > $L
for T in ml nj bigdeep; do
  timeout 300 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --opt walk_sort=0,1 2>&1 | grep -v "amdgpu.ids\|checksum" >> $L
done
cat $L
