#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/r03k_crown_sweep.log
: > $L
for KB in 256 768 1024 1536; do
  echo "=== SUCHTREE_AMD_CROWN_KB=$KB" >> $L
  for T in ml nj bigdeep; do
    SUCHTREE_AMD_CROWN_KB=$KB timeout 300 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --rounds 4 2>&1 | grep -v "amdgpu.ids\|checksum\|n_nodes" >> $L
  done
done
cat $L
( time python -m pytest tests -x -q -m gpu -k "parity or facade or host_path" ) > gpurun_out/r03k_pytest.log 2>&1
tail -5 gpurun_out/r03k_pytest.log
