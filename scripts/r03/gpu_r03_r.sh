#!/bin/bash
# tile size of k_walk_sorted by batch size (ml.tree, walk family forced; device-resident pairs)
set -u
mkdir -p gpurun_out
L=gpurun_out/r03r_walk_sort_q.log
: > $L
for N in 65536 100000 200000 400000 800000 1600000 3200000; do
  for Q in 1 2 4; do
    echo "== n=$N Q=$Q" >> $L
    timeout 120 python scripts/tune_gpu.py --tree ml --pairs $N --strategy walk --rounds 6 --opt sort_tile=$Q --opt walk_sort_min=1 --opt walk_sort=0,1 2>&1 | grep "median" >> $L
  done
done
cat $L
