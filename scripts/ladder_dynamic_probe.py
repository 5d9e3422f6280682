"""Scalar ladder kernel: static deal against work counters, by batch size (GPU box).
  python scripts/ladder_dynamic_probe.py ml nj s80 bigdeep"""
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
    nmax = 40_000_000
    pairs = torch.from_numpy(leaves[rng.integers(0, len(leaves), (nmax, 2))].astype(np.int64)).cuda()
    out_d = torch.empty(nmax, dtype=torch.float64, device="cuda")
    out_m = torch.empty(nmax, dtype=torch.int32, device="cuda")
    tree = _capi.DeviceTree(parent, dist)
    print(name, tree.info()["big_batch_kernel"], tree.info()["record_bytes"])
    base = dict(tile_sort=0, pairs_per_lane=1, ladder_scalar=1, ladder_min_pairs=0, prefer_walk_sorted=0)
    for k, v in base.items():
        tree.set_option(k, v)
    ref = None
    for n in (1 << 20, 1 << 21, 1 << 22, 10_000_000, 20_000_000, 40_000_000):
        line = "%10d " % n
        for dyn in (0, 1):
            tree.set_option("ladder_dynamic", 2 * dyn)      # (2: counters whatever the record size and the batch)
            for _ in range(2):
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            chk = (float(out_d[:n].sum()), int(out_m[:n].long().sum()))
            line += "  %s %.3f ms %.3e/s" % ("dynamic" if dyn else "static ", min(ts), n / min(ts) * 1e3)
            if dyn == 0:
                ref = chk
            else:
                line += "  same results: %s" % (chk == ref)
        print(line, flush=True)
    tree.close()
