# Final measurement suite of round 2 (GPU box): tests, configs, profiles, counters, bench.
mkdir -p gpurun_out/r2z
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r2z/pytest.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/r2z/pytest.log | tail -5
python scripts/bench_configs.py --configs 1,2,4,5,host,quartets --out gpurun_out/r2z/configs.jsonl > gpurun_out/r2z/configs.log 2>&1; cut -c1-260 gpurun_out/r2z/configs.jsonl
python scripts/latency_curve.py > gpurun_out/r2z/latency.log 2>&1
python scripts/host_path_probe.py > gpurun_out/r2z/host_probe.log 2>&1
python scripts/host_path_sweep.py > gpurun_out/r2z/host_sweep.jsonl 2>/dev/null
python scripts/bench_by_name.py > gpurun_out/r2z/by_name.json 2>/dev/null; cat gpurun_out/r2z/by_name.json
python scripts/pipe_trace.py > /dev/null 2> gpurun_out/r2z/pipe_trace.log
{ for t in ml nj; do echo "== $t canopy"; python scripts/tune_gpu.py --tree $t --pairs 10000000 --opt lineage_sums=0,1 2>&1 | grep -E "median"; echo "== $t walk"; python scripts/tune_gpu.py --tree $t --pairs 10000000 --strategy walk --opt lineage_sums=0,1 2>&1 | grep -E "median"; done; echo "== ml host path"; python scripts/ml_host_probe.py 2>&1 | grep "^ml"; } > gpurun_out/r2z/deep_lineage.log 2>&1; cat gpurun_out/r2z/deep_lineage.log
python scripts/deep_phase_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r2z/deep_phases.log; cat gpurun_out/r2z/deep_phases.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2z/trace_ml -- python3 $GRAFT_REPO_ROOT/scripts/tune_gpu.py --tree ml --pairs 10000000 --rounds 5 > /dev/null 2>&1 ); find gpurun_out/r2z/trace_ml -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2z/kernel_stats_ml.csv; head -3 gpurun_out/r2z/kernel_stats_ml.csv | cut -c1-220
bash scripts/profile_fabric.sh ml_final --tree ml --pairs 20000000 --rounds 3 > gpurun_out/r2z/fabric_ml.txt 2>&1
bash scripts/profile_counters.sh ml_sums --tree ml --pairs 20000000 --rounds 3 > gpurun_out/r2z/sq_ml_sums.txt 2>&1
bash scripts/profile_gpu.sh r02b > gpurun_out/r2z/profile.log 2>&1; cp profiles/*r02b* gpurun_out/r2z/
python bench.py > gpurun_out/r2z/bench.json 2> gpurun_out/r2z/bench.err; cut -c1-400 gpurun_out/r2z/bench.json
