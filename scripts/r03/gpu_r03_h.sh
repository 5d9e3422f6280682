#!/bin/bash
# walk family: crown (shared portal blocks + crown sparse table) on/off, by hot-set budget
set -u
mkdir -p gpurun_out
L=gpurun_out/r03h_walk_crown.log
: > $L
for KB in 2560 512 16384; do
  echo "=== SUCHTREE_AMD_CROWN_KB=$KB" >> $L
  for T in ml nj; do
    SUCHTREE_AMD_CROWN_KB=$KB timeout 300 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --opt walk_crown=0,1 2>&1 | grep -v "amdgpu.ids\|checksum" >> $L
  done
  SUCHTREE_AMD_CROWN_KB=$KB timeout 600 python scripts/big_deep_tree_probe.py 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
