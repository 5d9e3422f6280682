MATRIX_LOG2_MIN=14 MATRIX_LOG2_MAX=23 timeout 1200 python scripts/kernel_win_matrix.py r06deep cat2048 3000@0.97 8000@0.9 20000@0.95 2>&1 | grep -v Warning | tee gpurun_out/win_matrix_deep_r06.log
cp profiles/kernel_win_matrix_r06deep.json gpurun_out/ 2>/dev/null
