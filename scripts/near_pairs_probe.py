"""Pairs of nearby leaves (the MRCA lies below the canopy: both nodes share a portal) against uniform random
leaf pairs, by tree (GPU box).  2e7 device-resident pairs, distance + MRCA id, sample checked against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import OracleTree   # noqa: E402
from suchtree_amd import _capi, synth   # noqa: E402

n = 20_000_000
trees = {"balanced 2^20": synth.balanced_tree(20), "random shape 2^20": synth.random_binary_tree(1 << 20, seed=3)}
for name in ("ml", "nj"):
    z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
    trees[name + ".tree"] = (z["parent"], z["distance"])
for name, (parent, dist) in trees.items():
    tree = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    leaves = torch.from_numpy(np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0).astype(np.int64)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    ia = torch.randint(0, len(leaves), (n,), generator=g, device="cuda")
    out_d = torch.empty(n, dtype=torch.float64, device="cuda")
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    for label, spread in (("uniform", 0), ("within 1024 leaves", 1024), ("within 64 leaves", 64), ("within 8 leaves", 8)):
        if spread:
            ib = torch.clamp(ia + torch.randint(-spread, spread + 1, (n,), generator=g, device="cuda"), 0, len(leaves) - 1)
        else:
            ib = torch.randint(0, len(leaves), (n,), generator=g, device="cuda")
        pairs = torch.stack([leaves[ia], leaves[ib]], 1).contiguous()
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
        print("%-18s %-20s %7.2f ms  %.3e pairs/s  parity %s" % (name, label, min(ts), n / min(ts) * 1e3, "ok" if ok else "MISMATCH"), flush=True)
    tree.close()
