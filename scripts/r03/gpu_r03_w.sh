#!/bin/bash
set -u
( time python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_sorted_walk or walk_only_tree or walk_tables" ) > gpurun_out/r03w_pytest.log 2>&1
tail -5 gpurun_out/r03w_pytest.log
for T in ml nj bigdeep; do timeout 200 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --rounds 4 2>&1 | grep median; done
python scripts/big_deep_tree_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/big_deep_tree_r03.log | tail -7
