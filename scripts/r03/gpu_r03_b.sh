#!/bin/bash
set -u
mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_callers.py -x -q -m gpu -k "full_triangle" ) > gpurun_out/r03b_pytest.log 2>&1
echo "pytest rc=$?"
tail -25 gpurun_out/r03b_pytest.log
