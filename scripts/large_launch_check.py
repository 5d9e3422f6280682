"""One launch of more than 2^31 pairs (explicit int64 pair array in HBM, 2.3e9 pairs = 37 GB in,
28 GB out) on the headline tree and on ml.tree: 64-bit indexing end to end.  Checked against the
oracle on windows at the start, across the 2^31 and 2^32-byte boundaries and at the very end, and
against a second launch over the tail alone (GPU box, ~70 GB of HBM)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleTree   # noqa: E402
from suchtree_amd import _capi, synth   # noqa: E402

n = 2_300_000_001
dev = torch.device("cuda", 0)
z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
for name, (parent, dist, leaf_ids) in (("balanced20", synth.balanced_tree(20) + (np.arange(0, 1 << 21, 2, dtype=np.int64),)),
                                       ("ml", (z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)))):
    tree = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    leaves_t = torch.from_numpy(leaf_ids).to(dev)
    pairs = torch.empty((n, 2), dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    step = 200_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        idx = torch.randint(0, len(leaf_ids), (hi - lo, 2), generator=gen, device=dev)
        pairs[lo:hi] = leaves_t[idx]
        del idx
    out_d = torch.full((n,), -1.0, dtype=torch.float64, device=dev)
    out_m = torch.full((n,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tree.fault_check()
    ok = True
    for lo in (0, (1 << 31) // 16 - 5000, (1 << 31) - 5000, (1 << 32) // 8 - 5000, (1 << 32) // 4 - 5000, n - 10000):
        hi = min(n, lo + 10000)
        p = pairs[lo:hi].cpu().numpy()
        ok = ok and np.array_equal(out_d[lo:hi].cpu().numpy().view(np.int64), O.distances(p).view(np.int64))
        ok = ok and np.array_equal(out_m[lo:hi].cpu().numpy(), O.mrca_bulk(p))
    tail = 50_000_000
    t_d = torch.empty(tail, dtype=torch.float64, device=dev)
    t_m = torch.empty(tail, dtype=torch.int32, device=dev)
    tree.distances_device(pairs.data_ptr() + (n - tail) * 16, tail, t_d.data_ptr(), t_m.data_ptr())
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(t_d, out_d[n - tail:])) and bool(torch.equal(t_m, out_m[n - tail:]))
    ok = ok and not bool((out_m == -7).any().item())
    # MRCA ids alone over the whole batch
    out_m2 = torch.full((n,), -7, dtype=torch.int32, device=dev)
    tree.distances_device(pairs.data_ptr(), n, 0, out_m2.data_ptr())
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(out_m2, out_m))
    print("%s: %d pairs in one launch, %.3f s (%.3e pairs/s), parity %s" % (name, n, dt, n / dt, "ok" if ok else "MISMATCH"), flush=True)
    del pairs, out_d, out_m, out_m2, t_d, t_m
    tree.close()
    torch.cuda.empty_cache()
