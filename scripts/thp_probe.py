#!/usr/bin/env python3
"""What fresh result arrays cost on the GPU box, outside this library's control: transparent
huge page settings, the time of one host-path call into freshly allocated arrays, and the
time the caller then spends FREEING those arrays (munmap of 600 MB: ~20 ms, as much as the
call itself) -- compare with numpy's own first touch / free of a 400 MB array.
    python scripts/thp_probe.py          (from the repo root, on the GPU box)
"""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from suchtree_amd import _capi, synth
def huge():
    for l in open("/proc/self/smaps_rollup"):
        if l.startswith("AnonHugePages"): return l.split()[1] + " kB"
print("defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip(), "| shmem:", open("/sys/kernel/mm/transparent_hugepage/shmem_enabled").read().strip())
print({k: v for k, v in (l.split(":") for l in open("/proc/meminfo") if l.split(":")[0] in ("MemFree", "AnonHugePages", "HugePages_Total", "MemAvailable"))})
parent, dist = synth.balanced_tree(20)
tree = _capi.DeviceTree(parent, dist)
n = 50_000_000
pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
tree.distances_host(pairs[:5_000_000], True, True)
print("before:", huge())
t0 = time.perf_counter(); r = tree.distances_host(pairs, True, True); t1 = time.perf_counter()
print("fresh call %.1f ms -> AnonHugePages %s" % ((t1 - t0) * 1e3, huge()))
t0 = time.perf_counter(); del r; t1 = time.perf_counter()
print("free %.1f ms" % ((t1 - t0) * 1e3))
a = np.empty(n); t0 = time.perf_counter(); a[:] = 1.0; t1 = time.perf_counter()
print("numpy first touch of 400 MB: %.1f ms -> AnonHugePages %s" % ((t1 - t0) * 1e3, huge()))
t0 = time.perf_counter(); del a; t1 = time.perf_counter(); print("free %.1f ms" % ((t1 - t0) * 1e3))
