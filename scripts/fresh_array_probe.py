"""Fresh result arrays on the host path (GPU box): the first touch of the result pages by MADV_POPULATE_WRITE, by one
locked OR per page, or left to the unpack passes' own page faults -- alternating in ONE process chain on one box
(the pool's hosts differ by 2x).   python scripts/fresh_array_probe.py [pairs]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n):
    from suchtree_amd import _capi, synth
    os.environ["SUCHTREE_AMD_RECYCLE_MB"] = "0"      # every call allocates fresh numpy arrays
    parent, dist = synth.balanced_tree(20)
    tree = _capi.DeviceTree(parent, dist)
    pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
    h_d, h_m = np.empty(n), np.empty(n, np.int32)
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
    t0 = time.perf_counter()
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
    reused = time.perf_counter() - t0
    fresh = []
    for _ in range(4):
        t0 = time.perf_counter()
        r = tree.distances_host(pairs, True, True)
        fresh.append(time.perf_counter() - t0)
        del r
    print("reused %.3e  fresh %s pairs/s" % (n / reused, " ".join("%.3e" % (n / t) for t in fresh)))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50_000_000
    if "--child" in sys.argv:
        child(n)
    else:
        for rep in range(2):
            for mode in ("madv", "touch", "none", "touch+async", "madv+async"):
                env = dict(os.environ, SUCHTREE_AMD_POPULATE=mode.split("+")[0], SUCHTREE_AMD_TRACE_PIPE="1")
                if "async" in mode:
                    env["SUCHTREE_AMD_ASYNC_PREFAULT"] = "1"
                out = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "--child"], env=env, capture_output=True, text=True)
                print("%-11s %s" % (mode, out.stdout.strip()))
                print("      ", "\n       ".join(out.stderr.strip().splitlines()[-3:]))
