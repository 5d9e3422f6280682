#!/bin/bash
# Runs ON THE GPU BOX: the round's per-kernel profiles -- rocprofv3 kernel trace + PMC passes (scripts/profile_gpu.sh for the
# headline launch of bench.py, scripts/profile_kernel.sh for the kernels the bench's other legs run) -> profiles/kernel_stats_*_<round>.csv
# and profiles/traffic_*_<round>.json (copied back through gpurun_out/round_profiles/).
# usage: scripts/profile_round.sh r04
R=${1:-r04}
LADDER="--opt tile_sort=0 --opt ladder_scalar=1 --opt ladder_min_pairs=0"
bash scripts/profile_gpu.sh $R 2>&1 | tail -3
bash scripts/profile_kernel.sh ml_$R k_canopy_ladder 10000000 --tree ml $LADDER 2>&1 | tail -2
bash scripts/profile_kernel.sh nj_$R k_canopy_ladder 10000000 --tree nj $LADDER 2>&1 | tail -2
bash scripts/profile_kernel.sh s80_$R k_canopy_ladder 10000000 --tree shape:1000000:0.8 $LADDER 2>&1 | tail -2
bash scripts/profile_kernel.sh bigdeep_$R k_canopy_ladder 10000000 --tree bigdeep $LADDER 2>&1 | tail -2
bash scripts/profile_kernel.sh walk_bigdeep_$R k_walk_sorted 10000000 --tree bigdeep --strategy walk 2>&1 | tail -2
bash scripts/profile_kernel.sh walk_ml_$R k_walk_sorted 10000000 --tree ml --strategy walk 2>&1 | tail -2
mkdir -p gpurun_out/round_profiles
cp profiles/*_$R.csv profiles/*_$R.json gpurun_out/round_profiles/ 2>/dev/null
ls gpurun_out/round_profiles
