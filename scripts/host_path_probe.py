#!/usr/bin/env python3
"""Where the host path's time goes on short batches: fresh vs reused result arrays, by batch
size, for the headline tree and data/bigtrees/ml.tree (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi, synth   # noqa: E402


def best(fn, reps=5):
    fn()
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t = min(t, time.perf_counter() - t0)
    return t


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
    trees = {"ml": (z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)),
             "balanced20": synth.balanced_tree(20) + (np.arange(0, 1 << 21, 2, dtype=np.int64),)}
    for name, (parent, dist, leaves) in trees.items():
        tree = _capi.DeviceTree(parent, dist)
        for n in (1_000_000, 10_000_000, 50_000_000):
            pairs = np.random.default_rng(2).choice(leaves, size=(n, 2))
            d, m = np.empty(n), np.empty(n, np.int32)
            t_reused = best(lambda: tree.distances_host(pairs, True, True, out_dist=d, out_mrca=m))
            t_fresh = best(lambda: tree.distances_host(pairs, True, True))
            t_d = best(lambda: tree.distances_host(pairs, True, False, out_dist=d))
            print("%-10s n=%9d  reused %.3e  fresh %.3e  dist-only reused %.3e pairs/s" % (name, n, n / t_reused, n / t_fresh, n / t_d), flush=True)
        tree.close()


if __name__ == "__main__":
    main()
