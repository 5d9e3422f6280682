"""The generated pair sources on the deep trees (GPU box): the lower triangle over 10,000 random leaves (5e7 pairs,
device-resident results), the symmetric 5,000 x 5,000 matrix of pairwise_distances and the 16 nearest of 5,000
candidates for 5,000 queries through the host entry points."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import _capi, synth   # noqa: E402

trees = {"balanced 2^20": synth.balanced_tree(20)}
for name in ("ml", "nj"):
    z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
    trees[name + ".tree"] = (z["parent"], z["distance"])
for name, (parent, dist) in trees.items():
    tree = _capi.DeviceTree(parent, dist)
    leaves = np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0).astype(np.int64)
    rng = np.random.default_rng(3)
    ids = rng.choice(leaves, size=10_000, replace=False)
    m = len(ids)
    total = m * (m - 1) // 2
    d_ids = torch.from_numpy(ids).cuda()
    out_d = torch.empty(total, dtype=torch.float64, device="cuda")
    out_m = torch.empty(total, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.triangle_device(d_ids.data_ptr(), m, 0, total, out_d.data_ptr(), out_m.data_ptr())
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    i, j = np.tril_indices(m, -1)
    explicit = torch.from_numpy(np.stack([ids[j], ids[i]], 1)).cuda()
    chk_d = torch.empty(total, dtype=torch.float64, device="cuda")
    chk_m = torch.empty(total, dtype=torch.int32, device="cuda")
    te = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.distances_device(explicit.data_ptr(), total, chk_d.data_ptr(), chk_m.data_ptr())
        e1.record()
        e1.synchronize()
        te.append(e0.elapsed_time(e1))
    same = bool(torch.equal(out_d.view(torch.int64), chk_d.view(torch.int64)) and torch.equal(out_m, chk_m))
    print("%-14s triangle of %d leaves (%d pairs): generated %.2f ms = %.3e pairs/s; the same pairs as an explicit array %.2f ms = %.3e; identical %s"
          % (name, m, total, min(ts), total / min(ts) * 1e3, min(te), total / min(te) * 1e3, same), flush=True)
    sub = ids[:5000]
    tree.grid_host(sub, sub, symmetric=True)
    t0 = time.perf_counter()
    g, _ = tree.grid_host(sub, sub, symmetric=True)
    dt = time.perf_counter() - t0
    print("%-14s pairwise matrix %d x %d into host memory: %.1f ms = %.3e entries/s" % (name, len(sub), len(sub), dt * 1e3, len(sub) ** 2 / dt), flush=True)
    tree.knn_host(sub, sub, 16, skip_self=True)
    t0 = time.perf_counter()
    tree.knn_host(sub, sub, 16, skip_self=True)
    dt = time.perf_counter() - t0
    print("%-14s 16 nearest of %d candidates for %d queries: %.1f ms = %.3e candidate pairs/s" % (name, len(sub), len(sub), dt * 1e3, len(sub) ** 2 / dt), flush=True)
    tree.close()
