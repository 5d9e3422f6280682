#!/bin/bash
set -u
( time python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_sorted_walk or walk_only_tree or bigtrees" ) > gpurun_out/r03t_pytest.log 2>&1
tail -6 gpurun_out/r03t_pytest.log
python scripts/latency_curve.py 2>&1 | grep "^ml" | tail -6
