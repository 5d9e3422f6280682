"""bench.py prints exactly one JSON line with the contract's fields (small batch, GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--pairs", "4000000", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "pairs/s" and d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "pairs/s" and c["sample"]
    assert d["parity"]["distances_bit_exact"] and d["parity"]["mrca_bit_exact"]
    assert d["end_to_end_host_path"]["matches_device_results"]
    assert d["value"] > 1e9
    assert r["required_bytes_per_pair"] == 156 and 0 < r["required_frac"] < 1
    assert d["hardware_measured"]["table"]["Greads_per_s"] > 1 and d["hardware_measured"]["stream_copy_GBps"] > 100
    e = d["end_to_end_host_path"]
    assert e["pairs_per_s"] > 1e8 and e["pairs_per_s_fresh_arrays"] > 1e8 and e["pairs_per_s_call_and_drop_loop"] > 1e8
    assert d["mrca_ids_only"]["matches_the_fused_launch"] and d["mrca_ids_only"]["ids_per_s"] > 1e9


def test_bench_under_torchrun_one_rank():
    """The N > 1 code path (process group, sharded step, barriers) with world size 1 -- all a
    1-GPU box can run of it; the slicing / gather itself is covered by the gloo tests."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "2000000", "--no-cpu-baseline",
                          "--no-host-path", "--no-microbench"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 1e8
