#!/bin/bash
# walk family: tile-sorted kernel (walk_sort=1) vs k_walk
set -u
mkdir -p gpurun_out
L=gpurun_out/r03f_walk_sorted.log
: > $L
for T in ml nj; do
  timeout 300 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --opt walk_sort=0,1 >> $L 2>&1
done
timeout 300 python scripts/tune_gpu.py --levels 20 --pairs 20000000 --strategy walk --opt walk_sort=0,1 >> $L 2>&1
timeout 600 python scripts/big_deep_tree_probe.py >> $L 2>&1
grep -v "amdgpu.ids" $L | tail -40
