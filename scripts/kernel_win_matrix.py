"""Kernel x tree x batch size (GPU box, round 6): every kernel family of the library forced by options on the trees of
the verdict's list, device-resident uniform random leaf pairs, batch sizes 2^17 ... 2^24; per cell the median of 5 launches,
the winner and its margin over the runner-up -> profiles/kernel_win_matrix_<round>.json.  What wins no cell by more than 5 %
is a candidate for removal.  (Round 6 ran it three times -- large batches, small batches, small deep trees -- and merged the
three files with the decision into profiles/kernel_win_matrix_r06.json.)
    python scripts/kernel_win_matrix.py r06 [tree ...]
trees: ml nj 1e6@173 (1e6 leaves, skew 0.8) 1e6@338 (skew 0.9) 1e5@423 (1e5 leaves, skew 0.95) 2^20 (balanced)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SUCHTREE_AMD_AUTOTUNE"] = "0"      # every kernel is forced below; the timing at creation would only add launches
from suchtree_amd import _capi, synth      # noqa: E402
from test_gpu_parity import _random_shape_tree      # noqa: E402

TREES = {"ml": None, "nj": None, "1e6@173": (1_000_000, 0.8), "1e6@338": (1_000_000, 0.9), "1e5@423": (100_000, 0.95), "2^20": "balanced",
         "rand2^20": "random", "3e5@311": (300_000, 0.9), "1e6@108": (1_000_000, 0.7),
         # small deep trees: everything fits the canopy, records are short (1-7 slots)
         "cat2048": "caterpillar", "3000@0.97": (3000, 0.97), "8000@0.9": (8000, 0.9), "20000@0.95": (20000, 0.95)}
# kernel -> (strategy, options); options the kernel does not have on a tree make the cell "n/a"
KERNELS = {
    "canopy_sorted": ("canopy", {"tile_sort": 1, "ladder_scalar": 0, "prefer_walk_sorted": 0}),
    "canopy_ilp": ("canopy", {"tile_sort": 0, "ladder_scalar": 0, "prefer_walk_sorted": 0}),
    "canopy_ladder": ("canopy", {"tile_sort": 0, "ladder_scalar": 1, "ladder_min_pairs": 0, "prefer_walk_sorted": 0, "ladder_sums": 0}),
    "canopy_ladder_sums": ("canopy", {"tile_sort": 0, "ladder_scalar": 1, "ladder_min_pairs": 0, "prefer_walk_sorted": 0, "ladder_sums": 1}),
    "walk_sorted": ("walk", {"walk_sort": 1, "walk_sort_min": 1}),
    "walk": ("walk", {"walk_sort": 0}),
}
SIZES = [1 << k for k in range(int(os.environ.get("MATRIX_LOG2_MIN", "17")), int(os.environ.get("MATRIX_LOG2_MAX", "24")) + 1)]


def load(name):
    spec = TREES[name]
    if spec is None:
        z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
        return z["parent"], z["distance"]
    if spec == "balanced":
        return synth.balanced_tree(20)
    if spec == "caterpillar":
        return synth.caterpillar_tree(2048)
    if spec == "random":      # 2^20 leaves of random shape: a shallow canopy with longer records (the predicated kernel's 15- / 31-slot forms)
        return synth.random_binary_tree(1 << 20, seed=1)
    return _random_shape_tree(np.random.default_rng(5), spec[0], spec[1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    names = sys.argv[2:] or list(TREES)
    out_file = os.path.join(ROOT, "profiles", "kernel_win_matrix_%s.json" % tag)
    result = json.load(open(out_file)) if os.path.exists(out_file) else {"sizes": SIZES, "kernels": list(KERNELS), "trees": {}}
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    for name in names:
        parent, dist = load(name)
        tree = _capi.DeviceTree(parent, dist)
        info = tree.info()
        tree.set_option("batch_probe", 0)
        leaves = np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0).astype(np.int64)
        li = torch.from_numpy(leaves).to(dev)
        g = torch.Generator(device=dev).manual_seed(1)
        nmax = SIZES[-1]
        pairs = li[torch.randint(0, len(leaves), (nmax, 2), generator=g, device=dev)]
        out_d = torch.empty(nmax, dtype=torch.float64, device=dev)
        out_m = torch.empty(nmax, dtype=torch.int32, device=dev)
        cells = {}
        sums = {}
        for kname, (strategy, opts) in KERNELS.items():
            try:
                tree.set_strategy(strategy)
                for k, v in opts.items():
                    tree.set_option(k, v)
            except Exception as e:      # noqa: BLE001 -- this tree has no such family
                cells[kname] = {"n/a": str(e)[:80]}
                continue
            row = {}
            for n in SIZES:
                ms = []
                try:
                    for r in range(6):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
                        e1.record(stream)
                        torch.cuda.synchronize()
                        if r:
                            ms.append(e0.elapsed_time(e1))
                except Exception as e:      # noqa: BLE001
                    row[str(n)] = None
                    continue
                row[str(n)] = float(np.median(ms))
                chk = (float(out_d[:n].sum().item()), int(out_m[:n].long().sum().item()))
                if sums.setdefault(n, chk) != chk:
                    raise SystemExit("%s: %s at n = %d gives another result than the kernels before it" % (name, kname, n))
            cells[kname] = row
        tree.fault_check(stream.cuda_stream)
        tree.close()
        # several option sets can land on the same kernel (a tree without the ladder image: "canopy_ladder" runs the predicated
        # kernel): rows with identical times within 1 % over all sizes are reported as aliases of the first
        winners = {}
        for n in SIZES:
            t = sorted((row[str(n)], k) for k, row in cells.items() if isinstance(row.get(str(n)), float))
            if not t:
                continue
            margin = (t[1][0] / t[0][0] - 1.0) if len(t) > 1 else None
            winners[str(n)] = {"winner": t[0][1], "ms": t[0][0], "pairs_per_s": n / t[0][0] * 1e3, "runner_up": t[1][1] if len(t) > 1 else None,
                               "margin": margin}
        result["trees"][name] = {"info": {k: info[k] for k in ("depth", "n_leaves", "canopy_nodes", "understory_max", "record_bytes", "lineage_entries")},
                                 "ms": cells, "winners": winners}
        print(name, {n: (w["winner"], round(w["margin"] or 0, 3)) for n, w in winners.items()}, flush=True)
        with open(out_file, "w") as fh:
            json.dump(result, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
