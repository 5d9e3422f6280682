#!/usr/bin/env python3
"""Per-call time of SuchTree.distances_bulk by batch size on a 2^17-leaf tree and on ml.tree
(GPU box): where the mailbox, the walk kernel, the canopy kernels and the staged pipe take over."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import SuchTree, synth   # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
for name, T in (("balanced17", SuchTree(synth.balanced_tree(17))), ("ml", SuchTree((z["parent"], z["distance"])))):
    T.to_device()
    leaves = np.asarray(T.leaf_node_ids, dtype=np.int64)
    rng = np.random.default_rng(1)
    for n in (1, 64, 1000, 2048, 2049, 4095, 4096, 8192, 8193, 10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000):
        pairs = rng.choice(leaves, size=(n, 2))
        reps = max(3, min(500, int(2e6 / n)))
        for _ in range(3):
            T.distances_bulk(pairs)
        t0 = time.perf_counter()
        for _ in range(reps):
            T.distances_bulk(pairs)
        dt = (time.perf_counter() - t0) / reps
        print("%-10s n=%8d  %9.1f us per call  %.3e pairs/s" % (name, n, dt * 1e6, n / dt), flush=True)
    T.close()
