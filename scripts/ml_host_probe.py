"""ml.tree host path (device staging + tile-sorted kernel) at 2e7 pairs, reused result arrays (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi   # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
tree = _capi.DeviceTree(z["parent"], z["distance"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
pairs = np.random.default_rng(2).choice(z["leaf_ids"].astype(np.int64), size=(n, 2))
d, m = np.empty(n), np.empty(n, np.int32)
for want_m in (True, False):
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        tree.distances_host(pairs, True, want_m, out_dist=d, out_mrca=m if want_m else None)
        best = min(best, time.perf_counter() - t0)
    print("ml n=%d mrca=%d  %.3e pairs/s" % (n, want_m, n / best), flush=True)
