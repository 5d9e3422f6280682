#!/bin/bash
mkdir -p gpurun_out scripts/micro/bin
hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/sector_pair scripts/micro/sector_pair.hip || exit 1
timeout 120 scripts/micro/bin/sector_pair | tee gpurun_out/sector_pair_r03.log
