#!/bin/bash
set -u
mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_sorted_walk or walk_tables or four_byte" ) > gpurun_out/r03o_pytest.log 2>&1
tail -25 gpurun_out/r03o_pytest.log
