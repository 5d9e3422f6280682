"""`python bench.py --gpus N` (N > 1) without a torch.distributed.run environment: the process becomes the
parent of the launcher BEFORE anything of torch or HIP is loaded, relays the one JSON line, propagates the
exit code and kills the job when its timeout expires.  CPU only: the launcher is a stand-in
(tests/helpers/launcher_stub.py) that reports what its parent looked like."""
import json
import os
import subprocess
import sys
import time

from conftest import ROOT

STUB = os.path.join(ROOT, "tests", "helpers", "launcher_stub.py")
BENCH = os.path.join(ROOT, "bench.py")


def run(args, mode="ok", timeout=120):
    env = dict(os.environ, SUCHTREE_AMD_BENCH_LAUNCHER=STUB, STUB_MODE=mode)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_self_launch_spawns_the_driver_command_from_a_process_without_hip():
    out = run(["--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert len(lines) == 1                      # the banner line of the child is not relayed
    d = json.loads(lines[0])
    assert d["parent_is_bench"] and d["parent_has_hip"] is False
    a = d["argv"]
    # the command of the task contract: --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <args>
    assert a[0] == "--nnodes=1" and a[1:3] == ["--nproc-per-node", "4"] and a[3:5] == ["--master-addr", "127.0.0.1"]
    assert a[5] == "--master-port" and 1024 < int(a[6]) < 65536 and a[7] == BENCH
    assert a[8:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert d["world_size_env"] is None and d["ipc_mode"] == "0"


def test_self_launch_propagates_failure():
    out = run(["--gpus", "2"], mode="fail")
    assert out.returncode == 7
    assert "exited with code 7" in out.stderr


def test_self_launch_kills_the_job_at_its_timeout():
    t0 = time.time()
    out = run(["--gpus", "2", "--launch-timeout", "2"], mode="sleep")
    assert out.returncode == 124 and time.time() - t0 < 60
    assert "killed" in out.stderr and out.stdout.strip() == ""


def test_one_gpu_run_does_not_self_launch():
    """--gpus 1 stays one plain process (the driver's N = 1 form); on this CPU box it stops at the GPU check."""
    out = run(["--gpus", "1"])
    assert out.returncode != 0 and "MI355X" in out.stderr and "RCCL version banner" not in out.stdout


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
