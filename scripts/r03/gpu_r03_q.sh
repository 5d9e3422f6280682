#!/bin/bash
set -u
export PMC_TIMEOUT=240
mkdir -p gpurun_out/r3z
rm -rf gpurun_out/prof_r03
bash scripts/profile_gpu.sh r03 > gpurun_out/r3z/profile_headline.log 2>&1; grep "rc=" gpurun_out/r3z/profile_headline.log | tr '\n' ' '
cp profiles/*_r03.* gpurun_out/r3z/ 2>/dev/null
