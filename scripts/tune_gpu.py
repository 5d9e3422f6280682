"""A/B timing of kernel variants in ONE process on the GPU box (interleaved,
several rounds), e.g.

    python scripts/tune_gpu.py --levels 20 --pairs 100000000 --opt rec_a4=0,1

Prints median kernel ms and pairs/s per setting and checks that every setting
produces identical outputs.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=20)
    ap.add_argument("--tree", default="balanced", help="balanced | ml | nj | random | caterpillar | bigdeep | shape:<leaves>:<skew>")
    ap.add_argument("--pairs", type=int, default=100_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], help="name=v1,v2,...")
    ap.add_argument("--strategy", default="canopy")
    ap.add_argument("--triangle", type=int, default=0,
                    help="M: pairs generated on the device, a --pairs tile from the middle of the lower triangle of the complete M-leaf tree (seed 44)")
    args = ap.parse_args()
    import torch
    from suchtree_amd import _capi, synth
    dev = torch.device("cuda", 0)
    if args.triangle:
        parent, dist = synth.complete_tree(args.triangle, seed=44)
        leaf_ids = np.arange(0, 2 * args.triangle, 2)
    elif args.tree == "balanced":
        parent, dist = synth.balanced_tree(args.levels)
        leaf_ids = np.arange(0, len(parent), 2)
    elif args.tree == "caterpillar":       # deep, small canopy: 2^levels leaves on one ladder
        parent, dist = synth.caterpillar_tree(1 << args.levels)
        leaf_ids = np.arange(0, len(parent), 2)
    elif args.tree == "bigdeep":        # 1,000,000 leaves, random shape with skew 0.9: depth 338, the canopy family refuses it
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_parity import _random_shape_tree
        parent, dist = _random_shape_tree(np.random.default_rng(5), 1_000_000, 0.9)
        leaf_ids = np.arange(0, len(parent), 2)
    elif args.tree.startswith("shape:"):   # tests/test_gpu_parity.py::_random_shape_tree, seed 5 (profiles/kernel_choice_r03.log)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_parity import _random_shape_tree
        parent, dist = _random_shape_tree(np.random.default_rng(5), int(args.tree.split(":")[1]), float(args.tree.split(":")[2]))
        leaf_ids = np.arange(0, len(parent), 2)
    elif args.tree == "random":
        parent, dist = synth.random_binary_tree(1 << args.levels, seed=1)
        leaf_ids = np.arange(0, len(parent), 2)
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % args.tree))
        parent, dist, leaf_ids = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
    tree = _capi.DeviceTree(parent, dist, strategy="auto")
    tree.set_strategy(args.strategy)
    print(tree.info())
    n = args.pairs
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    li = torch.from_numpy(np.ascontiguousarray(leaf_ids)).to(dev)
    pairs = li[torch.randint(0, len(leaf_ids), (n, 2), generator=g, device=dev)] if not args.triangle else None
    k_mid = (args.triangle * (args.triangle - 1) // 2) // 2
    out_d = torch.empty(n, dtype=torch.float64, device=dev)
    out_m = torch.empty(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    settings = [{}]
    for o in args.opt:
        name, vals = o.split("=")
        settings = [dict(s, **{name: int(v)}) for s in settings for v in vals.split(",")]
    times = {i: [] for i in range(len(settings))}
    ref = None
    for r in range(args.rounds + 1):
        for i, s in enumerate(settings):
            for k, v in s.items():
                tree.set_option(k, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if args.triangle:
                tree.triangle_device(li.data_ptr(), args.triangle, k_mid, n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
            else:
                tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            if r == 0:
                chk = (float(out_d.sum().item()), int(out_m.long().sum().item()))
                if ref is None:
                    ref = (out_d.clone(), out_m.clone())
                else:
                    assert torch.equal(ref[0].view(torch.int64), out_d.view(torch.int64)), s
                    assert torch.equal(ref[1], out_m), s
                print("setting", s, "checksum", chk)
            else:
                times[i].append(e0.elapsed_time(e1))
    tree.fault_check(stream.cuda_stream)
    for i, s in enumerate(settings):
        t = float(np.median(times[i]))
        print("%-40s median %.3f ms  min %.3f ms  %.3e pairs/s" % (s, t, min(times[i]), n / t * 1e3))


if __name__ == "__main__":
    main()
