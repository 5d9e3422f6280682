#!/bin/bash
# Runs ON THE GPU BOX (round 6): A/B of the scalar ladder kernel's joint form (option ladder_sums: a's side from the lineage
# sums, meeting node from rec_p + the 64-bit sparse table, b's record by chunks, one LDS climb per pair) against the kernel as it
# was, in ONE process per tree (scripts/tune_gpu.py: interleaved rounds, identical-output check).  (The round's other A/B -- the
# headline kernel with a 13-level canopy -- needed two experiment switches that are gone again: profiles/headline_13levels_r06.log.)
P=${1:-10000000}
export SUCHTREE_AMD_AUTOTUNE=0
for T in nj ml shape:1000000:0.8 bigdeep; do
  echo "== $T, $P pairs"
  timeout 600 python scripts/tune_gpu.py --tree $T --pairs $P --rounds 7 --opt tile_sort=0 --opt ladder_scalar=1 --opt ladder_min_pairs=0 --opt batch_probe=0 --opt ladder_sums=0,1 2>&1 | grep -v Warning | grep "canopy_nodes\|median\|Error\|error\|assert" | sed -e "s/.*'canopy_nodes': \([0-9]*\).*'record_bytes': \([0-9]*\).*/   canopy_nodes \1 record_bytes \2/"
done
