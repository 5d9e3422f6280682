#!/bin/bash
set -u
for ZC in 1 0; do
SUCHTREE_AMD_SORTED_ZERO_COPY=$ZC python - <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, ".")
from suchtree_amd import _capi
from oracle.oracle import OracleTree
for name in ("ml", "nj"):
    z = np.load("tests/golden/%s_tree.npz" % name)
    parent, dist, leaf = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
    tree = _capi.DeviceTree(parent, dist)
    if name == "nj": tree.set_option("walk_sort", 0)      # keep nj on its canopy kernel for this comparison
    O = OracleTree(parent, dist)
    for n in (300_000, 2_000_000, 20_000_000):
        pairs = np.random.default_rng(2).choice(leaf, size=(n, 2))
        h_d, h_m = np.empty(n), np.empty(n, np.int32)
        want = O.distances_mt(pairs[:200000], 64)
        tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m); best = min(best, time.perf_counter() - t0)
        ok = np.array_equal(h_d[:200000].view(np.int64), want.view(np.int64)) and np.array_equal(h_m[:200000], O.mrca_bulk(pairs[:200000]))
        print("zero_copy=%s %s n=%d: %.3e pairs/s (%.0f us) parity %s" % (os.environ["SUCHTREE_AMD_SORTED_ZERO_COPY"], name, n, n / best, best * 1e6, ok))
PY
done
