"""After the prune (GPU box, round 6): what a handle with DEFAULT options (kernels and forms chosen by its own timing at creation) does
on every tree of profiles/kernel_win_matrix_r06.json, batch size by batch size, against the best cell of the matrix taken BEFORE the
prune -- the check that nothing a tree needed went away and that the timing picks what the matrix says.
    python scripts/default_vs_matrix.py [tree ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ["SUCHTREE_AMD_TUNE_CACHE"] = "0"
from suchtree_amd import _capi      # noqa: E402
import kernel_win_matrix as M      # noqa: E402  (its tree loader; it sets SUCHTREE_AMD_AUTOTUNE=0 for itself)

os.environ.pop("SUCHTREE_AMD_AUTOTUNE", None)


def main():
    matrix = json.load(open(os.path.join(ROOT, "profiles", "kernel_win_matrix_r06.json")))
    best = {}
    for run in matrix["runs"].values():
        for t, v in run["trees"].items():
            for n, w in v["winners"].items():
                best.setdefault(t, {})[int(n)] = (w["ms"], w["winner"])
    names = sys.argv[1:] or sorted(best)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    worst = 0.0
    for name in names:
        parent, dist = M.load(name)
        tree = _capi.DeviceTree(parent, dist)
        info = tree.info()
        leaves = np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0).astype(np.int64)
        li = torch.from_numpy(leaves).to(dev)
        g = torch.Generator(device=dev).manual_seed(1)
        sizes = sorted(best[name])
        nmax = sizes[-1]
        pairs = li[torch.randint(0, len(leaves), (nmax, 2), generator=g, device=dev)]
        out_d = torch.empty(nmax, dtype=torch.float64, device=dev)
        out_m = torch.empty(nmax, dtype=torch.int32, device=dev)
        print("== %s: default = %s, tuned %d, ladder_sums %d (up to %d pairs), record_bytes %d" %
              (name, info["big_batch_kernel"], info["tuned"], info["ladder_sums"], info["ladder_sums_max_pairs"], info["record_bytes"]), flush=True)
        for n in sizes:
            ms = []
            for r in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                if r:
                    ms.append(e0.elapsed_time(e1))
            t = float(np.median(ms))
            b_ms, b_k = best[name][n]
            worst = max(worst, t / b_ms - 1.0)
            flag = "   <-- %.0f %% behind" % (100 * (t / b_ms - 1)) if t > 1.08 * b_ms else ""
            print("   %9d pairs  default %.4f ms  best of the matrix %.4f ms (%s)%s" % (n, t, b_ms, b_k, flag), flush=True)
        tree.fault_check(stream.cuda_stream)
        tree.close()
    print("worst cell: default %.0f %% behind the matrix's best" % (100 * worst))


if __name__ == "__main__":
    main()
