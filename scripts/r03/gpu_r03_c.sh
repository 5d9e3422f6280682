#!/bin/bash
# walk family: b's side as a stream of lineage lengths (lineage_lens=1) vs the stride-3 climb (0)
set -u
mkdir -p gpurun_out
L=gpurun_out/r03c_walk_lens.log
: > $L
for T in ml nj; do
  python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --opt lineage_lens=0,1 >> $L 2>&1
done
python scripts/big_deep_tree_probe.py >> $L 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "walk" >> $L 2>&1
tail -40 $L
