"""Stand-in for `python -m torch.distributed.run` in tests/test_bench_launch.py (CPU only): records how
bench.py's self-launch parent called it and what that parent had mapped at the time."""
import json
import os
import sys
import time

mode = os.environ.get("STUB_MODE", "ok")
if mode == "sleep":
    time.sleep(600)
ppid = os.getppid()
with open("/proc/%d/maps" % ppid) as fh:
    maps = fh.read()
with open("/proc/%d/cmdline" % ppid, "rb") as fh:
    parent_cmd = fh.read().split(b"\0")
loaded = sorted({os.path.basename(l.split()[-1]) for l in maps.splitlines() if "/" in l and ".so" in l})
print("RCCL version banner that is not JSON")
print(json.dumps({"argv": sys.argv[1:], "parent_has_hip": any("amdhip" in x or "libtorch" in x or "hsa-runtime" in x for x in loaded),
                  "parent_is_bench": any(p.endswith(b"bench.py") for p in parent_cmd),
                  "world_size_env": os.environ.get("WORLD_SIZE"), "ipc_mode": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
sys.exit(7 if mode == "fail" else 0)
