"""Host path of the headline workload with the pipe's phase trace on (GPU box):
    python scripts/host_path_trace.py [pairs]
prints the machine's NUMA layout and the GPU's node, then the best of four reused-array calls and the
pipe trace of the last one: twice as the scheduler places the process, once confined to the GPU's
NUMA node, once to the other node, and once without the CPU passes (SUCHTREE_AMD_PIPE_SKIP_CPU=1: what
the GPU / link side of the pipeline does alone)."""
import os
import subprocess
import sys
import time

import numpy as np

os.environ["SUCHTREE_AMD_TRACE_PIPE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:      # noqa: BLE001
        return "(%s)" % e


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    if "--child" not in sys.argv:
        print(sh("lscpu | grep -i 'model name\\|socket\\|numa\\|^cpu(s)\\|thread'"))
        print("gpu numa nodes:", sh("cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\\n' ' '"))
        print("affinity:", len(os.sched_getaffinity(0)), "cpus;", sh("grep -i 'Cpus_allowed_list\\|Mems_allowed_list' /proc/self/status"))
        print("numactl:", sh("which numactl"), "| node of each card:", sh("for c in /sys/class/drm/card*/device; do echo -n \"$(basename $(dirname $c)):$(cat $c/numa_node) \"; done"))
        print("visible GPU pci bus:", sh("python3 -c \"import torch; print(torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0),'pci_bus_id') else '')\" 2>/dev/null"),
              sh("rocm-smi --showbus 2>/dev/null | grep -i 'GPU\\[' | head -3"), sh("rocm-smi --showtoponuma 2>/dev/null | grep -i numa | head -4"))
        gpu_node = sh("rocm-smi --showtoponuma 2>/dev/null | grep -i 'Numa Node' | head -1 | grep -o '[0-9]*$'") or "0"
        other = "1" if gpu_node == "0" else "0"
        cpus = {k: sh("cat /sys/devices/system/node/node%s/cpulist" % k) for k in (gpu_node, other)}
        settings = [("", {}), ("", {}), ("", {"ST_TRACE_WIRE24": "0"}),
                    ("taskset -c " + cpus[gpu_node], {}), ("taskset -c " + cpus[other], {}),
                    ("", {"SUCHTREE_AMD_PIPE_SKIP_CPU": "1"}), ("", {"SUCHTREE_AMD_PIPE_SKIP_GPU": "1"}),
                    ("", {"SUCHTREE_AMD_H2D_ENGINE": "1"}), ("", {}), ("", {"SUCHTREE_AMD_H2D_ENGINE": "1"}), ("", {}),
                    ("", {"SUCHTREE_AMD_H2D_ENGINE": "1", "SUCHTREE_AMD_PIPE_SKIP_CPU": "1"})]
        for prefix, extra in settings:
            env = dict(os.environ, **extra)
            cmd = (prefix.split() if prefix else []) + [sys.executable, os.path.abspath(__file__), str(n), "--child"]
            out = subprocess.run(cmd, env=env, capture_output=True, text=True)
            where = "process on the GPU's node" if cpus[gpu_node] in prefix and prefix else "process on the other node" if prefix else "process placed by the scheduler"
            note = {"SUCHTREE_AMD_PIPE_SKIP_CPU": "NO pack / unpack passes (link side alone, results not produced)",
                    "SUCHTREE_AMD_PIPE_SKIP_GPU": "NO launches (the CPU passes alone, results not produced)"}
            print("%-34s %s %s" % (where, " ".join(note.get(k, "%s=%s" % (k, v)) for k, v in extra.items()), out.stdout.strip()))
            print("   ", "\n    ".join(out.stderr.strip().splitlines()[-2:]))
        return
    from suchtree_amd import _capi, synth
    parent, dist = synth.balanced_tree(20)
    tree = _capi.DeviceTree(parent, dist)
    if os.environ.get("ST_TRACE_WIRE24") == "0":
        tree.set_option("wire24", 0)
    pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
    h_d, h_m = np.empty(n), np.empty(n, np.int32)
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
        best = min(best, time.perf_counter() - t0)
    p32 = pairs.astype(np.int32)
    tree.distances_host(p32, True, True, out_dist=h_d, out_mrca=h_m)
    b32 = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        tree.distances_host(p32, True, True, out_dist=h_d, out_mrca=h_m)
        b32 = min(b32, time.perf_counter() - t0)
    tree.distances_host(pairs, True, False, out_dist=h_d)
    bd = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        tree.distances_host(pairs, True, False, out_dist=h_d)
        bd = min(bd, time.perf_counter() - t0)
    t0 = time.perf_counter()
    fresh = tree.distances_host(pairs, True, True)
    tf = time.perf_counter() - t0
    del fresh
    i = tree.info()
    print("reused %.3e pairs/s (wire %d B in / %d B out)  int32 ids %.3e  distances only %.3e  fresh arrays %.3e"
          % (n / best, i["host_wire_bytes_in"], i["host_wire_bytes_out"], n / b32, n / bd, n / tf))
    # last trace line = the distances-only call before the fresh one; the both-output trace is the one before
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)


if __name__ == "__main__":
    main()
