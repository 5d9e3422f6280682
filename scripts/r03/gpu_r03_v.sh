#!/bin/bash
set -u
L=gpurun_out/r03v_ladder_sweep.log
: > $L
for T in ml nj bigdeep; do
for NODES in 2048 3072 4096 5120 6144 8192; do
for Q in 1 2 4; do
  echo -n "$T ladder_nodes=$NODES Q=$Q  " >> $L
  SUCHTREE_AMD_CROWN_LADDER_NODES=$NODES timeout 200 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --rounds 3 --opt sort_tile=$Q 2>&1 | grep "median" >> $L || echo "failed" >> $L
done; done; done
cat $L
