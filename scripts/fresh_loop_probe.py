"""`d = T.distances_bulk(pairs)` in a loop (result dropped every iteration), 1e7 and 5e7 pairs of the
headline workload: with the recycle pool (default) and without (SUCHTREE_AMD_RECYCLE_MB=0) (GPU box)."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from suchtree_amd import SuchTree, synth
    T = SuchTree(synth.balanced_tree(20)).to_device()
    for n in (10_000_000, 50_000_000):
        pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
        for what, fn in (("distances_bulk", lambda: T.distances_bulk(pairs)),
                         ("distances_and_ancestors_bulk", lambda: T.distances_and_ancestors_bulk(pairs))):
            fn()
            t0 = time.perf_counter()
            for _ in range(5):
                r = fn()
                del r
            dt = (time.perf_counter() - t0) / 5
            print("recycle_mb=%s  %-30s n=%d  %.3e pairs/s" % (os.environ.get("SUCHTREE_AMD_RECYCLE_MB", "default"), what, n, n / dt), flush=True)
else:
    for mb in ("0", None):
        env = dict(os.environ)
        if mb is not None:
            env["SUCHTREE_AMD_RECYCLE_MB"] = mb
        else:
            env.pop("SUCHTREE_AMD_RECYCLE_MB", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
