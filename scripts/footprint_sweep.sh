#!/bin/bash
# Runs ON THE GPU BOX: the product's headline kernel family (k_canopy_ilp, the kernel st_distances_device picks for a
# balanced tree) on balanced trees of 2^20, 2^22 and 2^24 leaves, 1e8 uniform random leaf pairs each -- record tables of
# 36 MiB (inside the 256 MiB Infinity Cache), ~290 MiB and ~1.1 GB gathered from (beyond it): rocprofv3 --kernel-trace
# --stats + one PMC pass per counter set (scripts/profile_kernel.sh), condensed into profiles/footprint_sweep_<round>.json.
# The TCC_EA0 counters sit at the L2's memory side and count Infinity-Cache hits too (MI355X_MICROARCH.md, HBM section), so
# the DRAM share is read off the footprint: at 2^24 leaves nothing the pairs touch stays resident.
# usage: scripts/footprint_sweep.sh r05 [levels...]
R=${1:-r05}; shift || true
LEVELS=${@:-"20 22 24"}
export PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
for L in $LEVELS; do
  bash scripts/profile_kernel.sh fp${L}_$R k_canopy_ilp 100000000 --tree balanced --levels $L 2>&1 | tail -3
done
python3 scripts/footprint_sweep_summary.py $R $LEVELS
mkdir -p gpurun_out/round_profiles
cp profiles/*fp*_$R.* profiles/footprint_sweep_$R.json gpurun_out/round_profiles/ 2>/dev/null
