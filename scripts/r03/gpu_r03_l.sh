#!/bin/bash
set -u
mkdir -p gpurun_out scripts/micro/bin
hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/gather_mix scripts/micro/gather_mix.hip || exit 1
timeout 300 scripts/micro/bin/gather_mix > gpurun_out/gather_mix_r03.log 2>&1
cat gpurun_out/gather_mix_r03.log
( time python -m pytest tests/test_gpu_callers.py -x -q -m gpu -k "special_values or neighbors" ) 2>&1 | tail -6
