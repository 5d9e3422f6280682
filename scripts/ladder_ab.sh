#!/bin/bash
# Runs ON THE GPU BOX: the scalar ladder kernel on the deep trees (default form of the current build), tune_gpu.py medians.
# usage: scripts/ladder_ab.sh [pairs] [trees...]
P=${1:-10000000}; shift || true
TREES=${@:-"nj ml shape:1000000:0.8 bigdeep"}
export SUCHTREE_AMD_AUTOTUNE=0
for T in $TREES; do
  echo "== $T, $P pairs"
  timeout 600 python scripts/tune_gpu.py --tree $T --pairs $P --rounds 7 --opt tile_sort=0 --opt ladder_scalar=1 --opt ladder_min_pairs=0 $EXTRA 2>&1 | grep -v Warning | grep "canopy_nodes\|median\|Error\|error\|assert" | sed -e "s/.*'canopy_nodes': \([0-9]*\).*'record_bytes': \([0-9]*\).*/   canopy_nodes \1 record_bytes \2/"
done
