#!/bin/bash
# Runs ON THE GPU BOX: the scalar ladder kernel on ml.tree / nj.tree with the deep canopy rebuilt at several sizes
# (SUCHTREE_AMD_DEEP_NODES: a smaller canopy = a shorter LDS climb, longer understory records added in registers).
# usage: scripts/deep_nodes_probe.sh [pairs]
P=${1:-10000000}
export SUCHTREE_AMD_AUTOTUNE=0
for T in ml nj; do
  for N in 10238 8192 6144 5120 4096 3072 2048 1024; do
    echo "== $T deep_nodes=$N"
    SUCHTREE_AMD_DEEP_NODES=$N timeout 300 python scripts/tune_gpu.py --tree $T --pairs $P --rounds 5 --opt tile_sort=0 --opt pairs_per_lane=1 --opt ladder_scalar=1 --opt ladder_min_pairs=0 2>&1 | grep -v Warning | grep "canopy_nodes\|median" | sed -e "s/.*'canopy_nodes': \([0-9]*\).*'record_bytes': \([0-9]*\).*/   canopy_nodes \1 record_bytes \2/"
  done
done
