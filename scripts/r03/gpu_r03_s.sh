#!/bin/bash
set -u
L=gpurun_out/r03s_walk_sort_q_big.log
: > $L
for T in ml bigdeep; do for N in 5000000 10000000 40000000; do for Q in 2 4; do
  echo "== $T n=$N Q=$Q" >> $L
  timeout 200 python scripts/tune_gpu.py --tree $T --pairs $N --strategy walk --rounds 4 --opt sort_tile=$Q --opt walk_sort=1 2>&1 | grep "median" >> $L
done; done; done
cat $L
