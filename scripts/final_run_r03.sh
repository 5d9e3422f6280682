#!/bin/bash
# Measurement suite of round 3 (GPU box): profiles of the four kernels that matter, ceilings sweep, bench line.
set -u
export PMC_TIMEOUT=240
mkdir -p gpurun_out/r3z
bash scripts/profile_gpu.sh r03 > gpurun_out/r3z/profile_headline.log 2>&1; tail -3 gpurun_out/r3z/profile_headline.log
bash scripts/profile_kernel.sh ml_r03 k_canopy_sorted 10000000 --tree ml > gpurun_out/r3z/profile_ml.log 2>&1; tail -2 gpurun_out/r3z/profile_ml.log
bash scripts/profile_kernel.sh walk_ml_r03 k_walk_sorted 10000000 --tree ml --strategy walk > gpurun_out/r3z/profile_walk_ml.log 2>&1; tail -2 gpurun_out/r3z/profile_walk_ml.log
bash scripts/profile_kernel.sh walk_bigdeep_r03 k_walk_sorted 10000000 --tree bigdeep --strategy walk > gpurun_out/r3z/profile_walk_bigdeep.log 2>&1; tail -2 gpurun_out/r3z/profile_walk_bigdeep.log
python3 - <<'PY' > gpurun_out/r3z/ceilings_sweep_r03.json 2> gpurun_out/r3z/ceilings.err
import json, sys
sys.path.insert(0, ".")
import bench_legs
log = []
out = bench_legs.hardware_ceilings(0, (1 << 20) * 20, sweep_log=log)
print(json.dumps({"best": out, "sweep": log}, indent=0))
PY
python scripts/big_deep_tree_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3z/big_deep_tree_r03.log; tail -7 gpurun_out/r3z/big_deep_tree_r03.log
python scripts/latency_curve.py > gpurun_out/r3z/latency_r03.log 2>&1; tail -4 gpurun_out/r3z/latency_r03.log
python bench.py > gpurun_out/r3z/bench_r03_selfrun.json 2> gpurun_out/r3z/bench.err; cut -c1-600 gpurun_out/r3z/bench_r03_selfrun.json
