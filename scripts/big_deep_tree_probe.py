"""A large, deep tree that the canopy family refuses (1,000,000 leaves, random shape with skew 0.9:
depth ~ several hundred, more than 16384 nodes above any 63-node understory): the walk family with
and without the whole-tree sparse table (GPU box)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import OracleTree   # noqa: E402
from suchtree_amd import _capi   # noqa: E402
from test_gpu_parity import _random_shape_tree   # noqa: E402

leaves, skew = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
parent, dist = _random_shape_tree(np.random.default_rng(5), leaves, skew)
t0 = time.perf_counter()
tree = _capi.DeviceTree(parent, dist)
print("create %.2f s" % (time.perf_counter() - t0), tree.info(), flush=True)
n = 10_000_000
rng = np.random.default_rng(2)
pairs_h = rng.integers(0, leaves, (n, 2)) * 2
pairs = torch.from_numpy(pairs_h).cuda()
out_d = torch.empty(n, dtype=torch.float64, device="cuda")
out_m = torch.empty(n, dtype=torch.int32, device="cuda")
O = OracleTree(parent, dist)
for on, lens, srt, crown, lad in ((1, 1, 1, 1, 1), (1, 1, 1, 1, 0), (1, 1, 0, 1, 0), (1, 1, 1, 0, 0), (1, 1, 0, 0, 0), (1, 0, 0, 0, 0), (0, 0, 0, 0, 0)):
    tree.set_option("tree_rmq", on)
    tree.set_option("lineage_lens", lens)
    tree.set_option("walk_sort", srt)
    tree.set_option("walk_crown", crown)
    tree.set_option("walk_ladder", lad)
    times = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    tree.fault_check()
    k = 20000
    ok = (np.array_equal(out_d[:k].cpu().numpy().view(np.int64), O.distances(pairs_h[:k]).view(np.int64))
          and np.array_equal(out_m[:k].cpu().numpy(), O.mrca_bulk(pairs_h[:k])))
    print("tree_rmq=%d lineage_lens=%d walk_sort=%d walk_crown=%d walk_ladder=%d  median %.3f ms  %.3e pairs/s  parity %s" % (on, lens, srt, crown, lad, float(np.median(times)), n / np.median(times) * 1e3, "ok" if ok else "MISMATCH"), flush=True)
