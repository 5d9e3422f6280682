"""Deep kernel on ml.tree / nj.tree, 1e7 leaf pairs in HBM: both outputs, MRCA ids only (= the key
phase of the lineage-sum mode), distances only (GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi   # noqa: E402

n = 10_000_000
from suchtree_amd import synth   # noqa: E402

for name in ("ml", "nj", "balanced20"):
    if name == "balanced20":
        parent, dist = synth.balanced_tree(20)
        leaf_ids = np.arange(0, len(parent), 2, dtype=np.int64)
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
        parent, dist, leaf_ids = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
    tree = _capi.DeviceTree(parent, dist)
    pairs = torch.from_numpy(np.random.default_rng(2).choice(leaf_ids, size=(n, 2))).cuda()
    out_d = torch.empty(n, dtype=torch.float64, device="cuda")
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    for label, pd, pm in (("both", out_d.data_ptr(), out_m.data_ptr()), ("mrca only", 0, out_m.data_ptr()), ("dist only", out_d.data_ptr(), 0)):
        times = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tree.distances_device(pairs.data_ptr(), n, pd, pm)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        print("%s %-10s median %.3f ms  %.3e pairs/s" % (name, label, float(np.median(times)), n / np.median(times) * 1e3), flush=True)
    tree.close()
