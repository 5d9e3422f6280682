"""Small trees, large batches (GPU box): everything but the pair and result streams is cache resident, so the
stream rate (16 B in + 12 B out per pair) is the ceiling.  1e8 device-resident pairs, sample checked."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleTree   # noqa: E402
from suchtree_amd import _capi, synth   # noqa: E402

n = 100_000_000
z = np.load(os.path.join(ROOT, "tests", "golden", "gopher_all_pairs.npz"))
trees = {"gopher (15 leaves)": (z["parent"], z["distance"])}
for leaves in (100, 1000, 10_000, 100_000):
    trees["random shape, %d leaves" % leaves] = synth.random_binary_tree(leaves, seed=4)
trees["caterpillar, 2000 leaves"] = synth.caterpillar_tree(2000)
for name, (parent, dist) in trees.items():
    tree = _capi.DeviceTree(parent, dist)
    info = tree.info()
    O = OracleTree(parent, dist)
    g = torch.Generator(device="cuda").manual_seed(1)
    for label, pairs in (("leaf pairs", None), ("node pairs", torch.randint(0, len(parent), (n, 2), generator=g, device="cuda"))):
        if pairs is None:
            leaves = torch.from_numpy(np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0).astype(np.int64)).cuda()
            pairs = leaves[torch.randint(0, len(leaves), (n, 2), generator=g, device="cuda")]
        out_d = torch.empty(n, dtype=torch.float64, device="cuda")
        out_m = torch.empty(n, dtype=torch.int32, device="cuda")
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        tree.fault_check()
        k = 20000
        ph = pairs[:k].cpu().numpy()
        ok = (np.array_equal(out_d[:k].cpu().numpy().view(np.int64), O.distances(ph).view(np.int64))
              and np.array_equal(out_m[:k].cpu().numpy(), O.mrca_bulk(ph)))
        print("%-28s %-7s depth %5d  %-10s %7.2f ms  %.3e pairs/s = %.2f TB/s of streams  parity %s"
              % (name, info["strategy"], info["depth"], label, min(ts), n / min(ts) * 1e3, 28.0 * n / min(ts) * 1e3 / 1e12, "ok" if ok else "MISMATCH"), flush=True)
        del pairs, out_d, out_m
    tree.close()
