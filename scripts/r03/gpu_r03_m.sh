#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/r03m_rec_a4.log
: > $L
timeout 300 python scripts/tune_gpu.py --levels 20 --pairs 100000000 --opt rec_a4=0,1 2>&1 | grep -v "amdgpu.ids" >> $L
timeout 300 python scripts/tune_gpu.py --levels 17 --pairs 20000000 --opt rec_a4=0,1 2>&1 | grep -v "amdgpu.ids" >> $L
timeout 300 python scripts/tune_gpu.py --tree random --levels 18 --pairs 20000000 --opt rec_a4=0,1 2>&1 | grep -v "amdgpu.ids" >> $L
cat $L
