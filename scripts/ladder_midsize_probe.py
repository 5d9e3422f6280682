"""Where the scalar ladder kernel takes over from the walk kernel (GPU box): device-resident batches of 2^12 .. 2^21
pairs, microseconds per call, canopy family with ladder_scalar = 1 / tile-sorted canopy kernel / k_walk.
  python scripts/ladder_midsize_probe.py ml nj s80 bigdeep"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi, synth   # noqa: E402

for name in sys.argv[1:] or ("ml", "nj"):
    if name in ("ml", "nj"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
        parent, dist = z["parent"], z["distance"]
    else:
        parent, dist = synth.skewed_tree(np.random.default_rng(5), 1_000_000, {"s80": 0.8, "bigdeep": 0.9}[name])
    leaves = np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0)
    rng = np.random.default_rng(3)
    nmax = 1 << 21
    pairs = torch.from_numpy(leaves[rng.integers(0, len(leaves), (nmax, 2))].astype(np.int64)).cuda()
    out_d = torch.empty(nmax, dtype=torch.float64, device="cuda")
    out_m = torch.empty(nmax, dtype=torch.int32, device="cuda")
    tree = _capi.DeviceTree(parent, dist)
    print(name, tree.info()["big_batch_kernel"], tree.info()["record_bytes"])
    cols = [("ladder", dict(tile_sort=0, pairs_per_lane=1, ladder_scalar=1, ladder_min_pairs=0, prefer_walk_sorted=0), "canopy"),
            ("sorted", dict(tile_sort=1, pairs_per_lane=0, ladder_scalar=0, prefer_walk_sorted=0), "canopy"),
            ("k_walk", dict(walk_sort_min=1 << 40), "walk"), ("walk_sorted", dict(walk_sort_min=32768), "walk")]
    print("%9s " % "pairs" + "".join("%14s" % c[0] for c in cols))
    for sh in range(12, 22):
        n = 1 << sh
        line = "%9d " % n
        for label, opts, strategy in cols:
            try:
                tree.set_strategy(strategy)
            except Exception:
                line += "%14s" % "-"
                continue
            for k, v in opts.items():
                tree.set_option(k, v)
            for _ in range(3):
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            e1.record()
            e1.synchronize()
            line += "%14.1f" % (e0.elapsed_time(e1) * 1e3 / reps)
        print(line, flush=True)
    tree.close()
