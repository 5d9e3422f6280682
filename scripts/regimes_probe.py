"""Query regimes beyond uniform random leaf pairs (GPU box): leaves within 8 of each other (shared portal), uniform
random NODE pairs (half of them internal), each with both outputs, MRCA ids only and distances only; samples checked
against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.oracle import OracleTree
from suchtree_amd import _capi, synth
n = 20_000_000
trees = {"balanced 2^20": synth.balanced_tree(20), "random 2^22": synth.random_binary_tree(1 << 22, seed=5)}
z = np.load("tests/golden/ml_tree.npz"); trees["ml.tree"] = (z["parent"], z["distance"])
for name, (parent, dist) in trees.items():
    tree = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    leaves = torch.from_numpy(np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0).astype(np.int64)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    ia = torch.randint(0, len(leaves), (n,), generator=g, device="cuda")
    out_d = torch.empty(n, dtype=torch.float64, device="cuda")
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    work = {}
    work["uniform leaves"] = torch.stack([leaves[ia], leaves[torch.randint(0, len(leaves), (n,), generator=g, device="cuda")]], 1).contiguous()
    work["leaves within 8"] = torch.stack([leaves[ia], leaves[torch.clamp(ia + torch.randint(-8, 9, (n,), generator=g, device="cuda"), 0, len(leaves) - 1)]], 1).contiguous()
    work["uniform nodes"] = torch.randint(0, len(parent), (n, 2), generator=g, device="cuda")
    for label, pairs in work.items():
        for what, dptr, mptr in (("dist+mrca", out_d.data_ptr(), out_m.data_ptr()), ("mrca only", 0, out_m.data_ptr()), ("dist only", out_d.data_ptr(), 0)):
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); tree.distances_device(pairs.data_ptr(), n, dptr, mptr); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            tree.fault_check()
            k = 20000
            ph = pairs[:k].cpu().numpy()
            ok = True
            if dptr: ok &= np.array_equal(out_d[:k].cpu().numpy().view(np.int64), O.distances(ph).view(np.int64))
            if mptr: ok &= np.array_equal(out_m[:k].cpu().numpy(), O.mrca_bulk(ph))
            print("%-14s %-16s %-10s %6.2f ms  %.3e /s  %s" % (name, label, what, min(ts), n / min(ts) * 1e3, "ok" if ok else "MISMATCH"), flush=True)
    tree.close()
