timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -60 | tee gpurun_out/tests_r06e.log
timeout 600 python tests/fuzz_parity.py 300 6001 2>&1 | tail -5 | tee gpurun_out/fuzz_r06.log
timeout 600 python tests/fuzz_parity.py 240 6002 big 2>&1 | tail -5 | tee -a gpurun_out/fuzz_r06.log
