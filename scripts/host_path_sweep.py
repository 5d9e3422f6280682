#!/usr/bin/env python3
"""Host-buffer path throughput (PCIe inclusive) on the headline tree, fresh vs reused result
arrays, for a given number of copy threads (SUCHTREE_AMD_COPY_THREADS).  Runs on the GPU box:
    for t in 8 16 32; do SUCHTREE_AMD_COPY_THREADS=$t python scripts/host_path_sweep.py; done
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suchtree_amd import _capi, synth   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    parent, dist = synth.balanced_tree(20)
    tree = _capi.DeviceTree(parent, dist)
    pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
    h_d, h_m = np.empty(n), np.empty(n, np.int32)
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
    out = {"threads": os.environ.get("SUCHTREE_AMD_COPY_THREADS", "default"), "pairs": n,
           "thp": open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()}
    for name, kw in (("reused", dict(out_dist=h_d, out_mrca=h_m)), ("fresh", {})):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = tree.distances_host(pairs, True, True, **kw)
            best = min(best, time.perf_counter() - t0)
            del r
        out[name + "_pairs_per_s"] = n / best
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = tree.distances_host(pairs, True, False)
        best = min(best, time.perf_counter() - t0)
        del r
    out["fresh_dist_only_pairs_per_s"] = n / best
    pooled = _capi.DeviceTree(parent, dist, pinned_results=True)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = pooled.distances_host(pairs, True, True)       # result arrays from the pinned pool, written directly
        best = min(best, time.perf_counter() - t0)
        del r
    out["pinned_pool_pairs_per_s"] = n / best
    t0 = time.perf_counter()
    for _ in range(3):
        r = tree.distances_host(pairs, True, True)
        del r                                              # the caller's side of a fresh array: freeing it
    out["fresh_incl_free_pairs_per_s"] = 3 * n / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(3):
        r = pooled.distances_host(pairs, True, True)
        del r
    out["pinned_pool_incl_free_pairs_per_s"] = 3 * n / (time.perf_counter() - t0)
    p32 = pairs.astype(np.int32)
    t0 = time.perf_counter()
    tree.distances_host(p32, True, True, out_dist=h_d, out_mrca=h_m)
    out["reused_int32_ids_pairs_per_s"] = n / (time.perf_counter() - t0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
