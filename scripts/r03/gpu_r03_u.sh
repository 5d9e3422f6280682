#!/bin/bash
set -u
L=gpurun_out/r03u_walk_ladder.log
: > $L
for T in ml nj bigdeep; do
  timeout 300 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --rounds 4 --opt walk_ladder=0,1 2>&1 | grep -v "amdgpu.ids\|checksum" >> $L
done
for NODES in 4096 6144; do
  echo "== SUCHTREE_AMD_CROWN_LADDER_NODES=$NODES" >> $L
  for T in ml bigdeep; do
    SUCHTREE_AMD_CROWN_LADDER_NODES=$NODES timeout 300 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --rounds 4 --opt walk_ladder=1 2>&1 | grep "median" >> $L
  done
done
cat $L
