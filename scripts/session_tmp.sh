timeout 1500 python scripts/default_vs_matrix.py 2>&1 | grep -v Warning | tee gpurun_out/default_vs_matrix_r06c.log | grep "behind\|==\|worst"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/tests_r06f.log
bash scripts/gpu_session.sh bench --steps 20 --warmup 5 > gpurun_out/bench_r06_d.txt
