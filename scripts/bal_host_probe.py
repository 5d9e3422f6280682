"""Headline tree host path at n pairs, reused result arrays (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi, synth   # noqa: E402

tree = _capi.DeviceTree(*synth.balanced_tree(20))
for n in [int(x) for x in (sys.argv[1:] or (1_000_000, 20_000_000))]:
    pairs = np.random.default_rng(2).integers(0, 1 << 20, size=(n, 2)) * 2
    d, m = np.empty(n), np.empty(n, np.int32)
    for want_m in (True, False):
        best = 1e9
        for _ in range(8):
            t0 = time.perf_counter()
            tree.distances_host(pairs, True, want_m, out_dist=d, out_mrca=m if want_m else None)
            best = min(best, time.perf_counter() - t0)
        print("bal n=%d mrca=%d  %.3e pairs/s  %.1f us" % (n, want_m, n / best, best * 1e6), flush=True)
