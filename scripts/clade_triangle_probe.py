"""All-pairs triangles over clades of a deep tree (consecutive leaves), device resident: the scalar ladder kernel
against the handle's other kernel (GPU box).   python scripts/clade_triangle_probe.py ml nj"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["SUCHTREE_AMD_AUTOTUNE"] = "0"
from suchtree_amd import _capi
for which in sys.argv[1:] or ("ml", "nj"):
    z = np.load("tests/golden/%s_tree.npz" % which); p, d = z["parent"], z["distance"]
    tree = _capi.DeviceTree(p, d)
    leaves = np.flatnonzero(np.bincount(p[p >= 0], minlength=len(p)) == 0).astype(np.int64)
    print(which, tree.info()["record_bytes"], flush=True)
    for m, start in ((2000, 1000), (4000, 20000), (8000, 10000), (len(leaves), 0)):
        ids = torch.from_numpy(leaves[start:start + m].copy()).cuda()
        m = len(ids)
        n = m * (m - 1) // 2
        if n > 400_000_000: n = 400_000_000
        out_d = torch.empty(n, dtype=torch.float64, device="cuda"); out_m = torch.empty(n, dtype=torch.int32, device="cuda")
        line = "   %6d leaves from %6d  %11d pairs " % (m, start, n)
        sums = []
        for label, opts in (("ladder", dict(tile_sort=0, pairs_per_lane=1, ladder_scalar=1, ladder_min_pairs=0)),
                            ("tile-sorted", dict(tile_sort=1, pairs_per_lane=0, ladder_scalar=0)),
                            ("predicated", dict(tile_sort=0, pairs_per_lane=1, ladder_scalar=0))):
            for k, v in opts.items(): tree.set_option(k, v)
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); tree.triangle_device(ids.data_ptr(), m, 0, n, out_d.data_ptr(), out_m.data_ptr()); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            sums.append((float(out_d.sum()), int(out_m.long().sum())))
            line += " %s %.2f ms %.2e/s " % (label, min(ts), n / min(ts) * 1e3)
        print(line, "same bits" if len(set(sums)) == 1 else "DIFFERENT", flush=True)
    tree.close()
