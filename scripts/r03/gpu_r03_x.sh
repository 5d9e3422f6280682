#!/bin/bash
set -u
export PMC_TIMEOUT=240
mkdir -p gpurun_out/r3z
bash scripts/profile_kernel.sh walk_ml_r03 k_walk_sorted 10000000 --tree ml --strategy walk > gpurun_out/r3z/profile_walk_ml.log 2>&1; tail -2 gpurun_out/r3z/profile_walk_ml.log
bash scripts/profile_kernel.sh walk_bigdeep_r03 k_walk_sorted 10000000 --tree bigdeep --strategy walk > gpurun_out/r3z/profile_walk_bigdeep.log 2>&1; tail -2 gpurun_out/r3z/profile_walk_bigdeep.log
