bash scripts/sector_ceiling_counters.sh 20 r06 2>&1 | tee gpurun_out/ceiling_counters_r06.log
bash scripts/profile_gpu.sh r06 2>&1 | tail -4
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/tests_r06d.log
bash scripts/gpu_session.sh bench --steps 20 --warmup 5 > gpurun_out/bench_r06_c.txt
