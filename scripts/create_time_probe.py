"""Time of st_tree_create (table construction on the host + upload) by tree (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from suchtree_amd import _capi, synth   # noqa: E402

trees = {"balanced 2^20": synth.balanced_tree(20), "complete 100k": synth.complete_tree(100_000, seed=44)}
z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
trees["ml.tree"] = (z["parent"], z["distance"])
from test_gpu_parity import _random_shape_tree   # noqa: E402
trees["1e6 leaves, depth 338"] = _random_shape_tree(np.random.default_rng(5), 1_000_000, 0.9)
_capi.DeviceTree(*trees["complete 100k"]).close()      # runtime start-up, pipe allocation
for name, (p, d) in trees.items():
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        t = _capi.DeviceTree(p, d)
        best = min(best, time.perf_counter() - t0)
        info = t.info()
        t.close()
    print("%-24s %8d nodes  create %.3f s  device tables %.1f MB" % (name, len(p), best, info["device_bytes"] / 1e6))
