#!/bin/bash
set -u
for T in nj ml; do
  timeout 200 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy canopy --rounds 4 --opt walk_sort=1,0 2>&1 | grep "median"
done
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from suchtree_amd import _capi
from oracle.oracle import OracleTree
for name in ("nj", "ml"):
    z = np.load("tests/golden/%s_tree.npz" % name)
    parent, dist, leaf = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
    tree = _capi.DeviceTree(parent, dist)
    n = 20_000_000
    pairs = np.random.default_rng(2).choice(leaf, size=(n, 2))
    h_d, h_m = np.empty(n), np.empty(n, np.int32)
    O = OracleTree(parent, dist)
    want = O.distances_mt(pairs[:300000], 64)
    for ws in (1, 0):
        tree.set_option("walk_sort", ws)
        tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m); best = min(best, time.perf_counter() - t0)
        ok = np.array_equal(h_d[:300000].view(np.int64), want.view(np.int64))
        print(name, "host path walk_sort=%d: %.3e pairs/s  parity %s" % (ws, n / best, ok))
PY
