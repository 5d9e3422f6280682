mkdir -p gpurun_out/r2k
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r2k/pytest.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/r2k/pytest.log | tail -5
python scripts/bench_configs.py --configs 1,2,4,5,3-host,f4 --out gpurun_out/r2k/configs.jsonl > gpurun_out/r2k/configs.log 2>&1; cut -c1-330 gpurun_out/r2k/configs.jsonl
bash scripts/profile_counters.sh r02_headline --levels 20 --pairs 100000000 --rounds 2 > gpurun_out/r2k/sq_headline.txt 2>&1
bash scripts/profile_counters.sh r02_ml --tree ml --pairs 20000000 --rounds 2 > gpurun_out/r2k/sq_ml.txt 2>&1
tail -25 gpurun_out/r2k/sq_headline.txt; tail -25 gpurun_out/r2k/sq_ml.txt
python bench.py > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err; cut -c1-600 gpurun_out/r2k/bench.json
