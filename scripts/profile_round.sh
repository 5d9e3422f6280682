#!/bin/bash
# Runs ON THE GPU BOX: the round's per-kernel profiles -- rocprofv3 kernel trace + PMC passes (scripts/profile_gpu.sh for the
# headline launch of bench.py, scripts/profile_kernel.sh for the kernels the bench's other legs run) -> profiles/kernel_stats_*_<round>.csv
# and profiles/traffic_*_<round>.json (copied back through gpurun_out/round_profiles/).  Every leg of bench.py's other_configs has
# its tag here: ml / nj (config 2, the kernel and form the handle times itself into: nj.tree the joint form, ml.tree the climbing
# form at 1e7 pairs), walk_ml / walk_nj, tri / walk_tri (config 4: pairs generated on the device, no id stream), s80, bigdeep,
# walk_bigdeep.
# usage: scripts/profile_round.sh r06
R=${1:-r06}
LADDER="--opt tile_sort=0 --opt ladder_scalar=1 --opt ladder_min_pairs=0 --opt batch_probe=0"
bash scripts/profile_gpu.sh $R 2>&1 | tail -3
bash scripts/profile_kernel.sh ml_$R k_canopy_ladder 10000000 --tree ml $LADDER --opt ladder_sums=0 2>&1 | tail -2
bash scripts/profile_kernel.sh nj_$R k_canopy_ladder 10000000 --tree nj $LADDER --opt ladder_sums=1 2>&1 | tail -2
bash scripts/profile_kernel.sh s80_$R k_canopy_ladder 10000000 --tree shape:1000000:0.8 $LADDER --opt ladder_sums=0 2>&1 | tail -2
bash scripts/profile_kernel.sh bigdeep_$R k_canopy_ladder 10000000 --tree bigdeep $LADDER 2>&1 | tail -2
bash scripts/profile_kernel.sh walk_bigdeep_$R k_walk_sorted 10000000 --tree bigdeep --strategy walk 2>&1 | tail -2
bash scripts/profile_kernel.sh walk_ml_$R k_walk_sorted 10000000 --tree ml --strategy walk 2>&1 | tail -2
bash scripts/profile_kernel.sh walk_nj_$R k_walk_sorted 10000000 --tree nj --strategy walk 2>&1 | tail -2
STREAM_BYTES_PER_PAIR=0 bash scripts/profile_kernel.sh tri_$R k_canopy_ilp 134217728 --triangle 100000 --strategy canopy 2>&1 | tail -2
STREAM_BYTES_PER_PAIR=0 bash scripts/profile_kernel.sh walk_tri_$R k_walk 134217728 --triangle 100000 --strategy walk 2>&1 | tail -2
mkdir -p gpurun_out/round_profiles
cp profiles/*_$R.csv profiles/*_$R.json gpurun_out/round_profiles/ 2>/dev/null
ls gpurun_out/round_profiles
