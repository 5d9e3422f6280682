#!/bin/bash
# round 3, first GPU visit: the restructured bench (contract tests + one default run)
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r03a_pytest.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/r03a_pytest.log
( time python bench.py > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err ) 2>&1 | tail -3
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r03a.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
print("hw", json.dumps(d.get("hardware_measured")))
print("rs", json.dumps(d.get("random_sector")))
print("host", json.dumps({k:v for k,v in d["end_to_end_host_path"].items() if k!="what"}))
for k,v in d["other_configs"].items(): print(k, json.dumps(v)[:900])
PY
