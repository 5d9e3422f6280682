#!/bin/bash
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from suchtree_amd import _capi, synth
from oracle.oracle import OracleTree
for name in ("bal20", "ml"):
    if name == "bal20":
        parent, dist = synth.balanced_tree(20); leaf = np.arange(0, len(parent), 2)
    else:
        z = np.load("tests/golden/ml_tree.npz"); parent, dist, leaf = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
    tree = _capi.DeviceTree(parent, dist)
    n = 100_000_000 if name == "bal20" else 10_000_000
    li = torch.from_numpy(np.ascontiguousarray(leaf)).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    pairs = li[torch.randint(0, len(leaf), (n, 2), generator=g, device="cuda")]
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    ms = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tree.distances_device(pairs.data_ptr(), n, 0, out_m.data_ptr()); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    tree.fault_check()
    k = 200000
    ok = np.array_equal(out_m[:k].cpu().numpy(), OracleTree(parent, dist).mrca_bulk(pairs[:k].cpu().numpy()))
    print(name, "MRCA ids only: %.3f ms  %.3e ids/s  parity %s" % (np.median(ms[1:]), n / np.median(ms[1:]) * 1e3, ok))
PY
