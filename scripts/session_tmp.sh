ls -la oracle/_ref/ 2>&1 | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warning | tail -3 | tee gpurun_out/smoke_r06b.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/tests_r06g.log
bash scripts/gpu_session.sh bench --steps 20 --warmup 5 > gpurun_out/bench_r06_e.txt
