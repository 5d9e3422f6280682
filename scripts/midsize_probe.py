"""Device-resident batches of 3e4 .. 4e6 pairs on the deep trees: kernel time by batch size, family and tile of
the tile-sorted kernels (sort_tile option: 0 = by batch size, the default; 1 / 2 / 4 = fixed) (GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi   # noqa: E402

for name in sys.argv[1:] or ("ml", "nj"):
    if name.startswith("balanced"):
        from suchtree_amd import synth
        parent, dist = synth.balanced_tree(int(name[8:]))
    elif name.startswith("random"):
        from suchtree_amd import synth
        parent, dist = synth.random_binary_tree(1 << int(name[6:]), seed=3)
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
        parent, dist = z["parent"], z["distance"]
    leaves = np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0)
    rng = np.random.default_rng(3)
    nmax = 1 << 22
    pairs = torch.from_numpy(leaves[rng.integers(0, len(leaves), (nmax, 2))].astype(np.int64)).cuda()
    out_d = torch.empty(nmax, dtype=torch.float64, device="cuda")
    out_m = torch.empty(nmax, dtype=torch.int32, device="cuda")
    print("%s: microseconds per call (both outputs), by batch size" % name)
    cols = [("auto", 0, 0), ("auto", 1, 0), ("auto", 2, 0), ("auto", 4, 0), ("walk", 0, 0), ("walk", 0, 1 << 40),
            ("walk", 1, 32768), ("walk", 2, 32768), ("walk", 4, 32768)]
    print("%9s " % "pairs" + "".join("%16s" % ("%s t=%d%s" % (s, q, " unsorted" if m > 1 << 30 else " min32k" if m else "")) for s, q, m in cols))
    trees = {}
    for strategy in ("auto", "walk"):
        trees[strategy] = _capi.DeviceTree(parent, dist)
        if strategy == "walk":
            trees[strategy].set_strategy("walk")
    for sh in range(int(os.environ.get('MIDSIZE_FIRST', '15')), 23):
        n = 1 << sh
        line = "%9d " % n
        for strategy, q, wmin in cols:
            tree = trees[strategy]
            tree.set_option("sort_tile", q)
            tree.set_option("walk_sort_min", wmin if strategy == "walk" else 1 << 40)   # (auto column: the canopy family's own kernel)
            for _ in range(3):
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            e1.record()
            e1.synchronize()
            line += "%16.1f" % (e0.elapsed_time(e1) * 1e3 / reps)
        print(line, flush=True)
    for t in trees.values():
        t.close()
