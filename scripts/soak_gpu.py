"""Soak test on the GPU box: random batch sizes / layouts / output selections / kernel options,
several host threads sharing handles, results checked against the oracle.  Not part of the
pytest suite (minutes long); run manually:  python scripts/soak_gpu.py --seconds 180"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--threads", type=int, default=4)
    args = ap.parse_args()
    from oracle.oracle import OracleTree
    from suchtree_amd import _capi, synth
    z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
    trees = [(z["parent"], z["distance"]), synth.balanced_tree(16), synth.random_binary_tree(30000, seed=3),
             synth.caterpillar_tree(2500)]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _random_shape_tree
    trees.append(_random_shape_tree(np.random.default_rng(7), 120_000, 0.97))      # only the walk family serves it
    trees.append(_random_shape_tree(np.random.default_rng(5), 200_000, 0.9))       # 1 KB records (no id chains)
    trees.append(synth.balanced_tree(15))                                          # under a table budget: no id chains
    devs = [_capi.DeviceTree(p, d, table_mb=(4.5 if i == 6 else None)) for i, (p, d) in enumerate(trees)]
    print([(x.info()["strategy"], x.info()["record_bytes"], x.info()["dropped_tables"]) for x in devs], flush=True)
    oracles = [OracleTree(p, d) for p, d in trees]
    stop = time.time() + args.seconds
    errors, counts = [], [0] * args.threads

    def worker(tid):
        rng = np.random.default_rng(1000 + tid)
        while time.time() < stop and not errors:
            k = int(rng.integers(0, len(devs)))
            dev, O, n_nodes = devs[k], oracles[k], len(trees[k][0])
            n = int(10 ** rng.uniform(0, 6.3))
            pairs = rng.integers(0, n_nodes, (n, 2))
            if rng.random() < 0.35:      # pairs of nearby nodes: one portal (ids follow from another phase / the walk fallback)
                a = rng.integers(0, max(1, n_nodes - 40), n)
                pairs = np.stack([a, np.minimum(n_nodes - 1, a + rng.integers(0, 40, n))], 1)
            layout = rng.integers(0, 4)
            if layout == 1:
                view = np.asfortranarray(pairs)
            elif layout == 2:
                wide = np.zeros((n, 5), dtype=np.int64)
                wide[:, 1::3] = pairs
                view = wide[:, 1::3]
            elif layout == 3:
                view = pairs.astype(np.int64)[::-1][::-1]
            else:
                view = pairs
            want_d, want_m = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            if not (want_d or want_m):
                want_d = True
            try:
                if rng.random() < 0.3:
                    dev.set_option("small_batch_path", int(rng.integers(0, 2)))
                    if k == 0:
                        dev.set_option("tile_sort", int(rng.integers(0, 2)))
                        dev.set_option("lineage_sums", int(rng.integers(0, 2)))
                    dev.set_option("mrca_ranks", int(rng.integers(0, 2)))
                    dev.set_option("sort_tile", int(rng.choice([0, 0, 1, 2, 4])))
                    dev.set_option("rec_a4", int(rng.integers(0, 2)))
                    dev.set_option("wire48", int(rng.integers(0, 2)))
                    dev.set_option("wire24", int(rng.random() < 0.8))
                    if k in (0, 2):
                        dev.set_option("ladder_scalar", int(rng.integers(0, 2)))
                        dev.set_option("ladder_min_pairs", int(rng.choice([0, 0, 300000])))
                        dev.set_option("ladder_dynamic", int(rng.choice([0, 1])))
                        dev.set_option("ladder_sums", int(rng.choice([0, 1])))      # (the joint form, round 6)
                        dev.set_option("batch_probe", int(rng.choice([0, 1])))
                    if k in (0, 4):
                        for name in ("walk_sort", "walk_ladder", "walk_crown", "lineage_lens"):
                            dev.set_option(name, int(rng.random() < 0.8))
                        dev.set_option("walk_sort_min", int(rng.choice([0, 40000])))
                        if k == 0:
                            dev.set_strategy("walk" if rng.random() < 0.3 else "canopy")
                if rng.random() < 0.2:
                    view = np.ascontiguousarray(view).astype(np.int32)      # the int32 entry point
                d, m = dev.distances_host(view, want_d, want_m)
                # the head, the tail and a random window of the batch
                for lo in {0, max(0, n - 20000), int(rng.integers(0, max(1, n - 20000)))}:
                    hi = min(n, lo + 20000)
                    if want_d and not np.array_equal(d[lo:hi].view(np.int64), O.distances(pairs[lo:hi]).view(np.int64)):
                        errors.append("dist mismatch tree %d n %d layout %d at %d" % (k, n, layout, lo))
                    if want_m and not np.array_equal(m[lo:hi], O.mrca_bulk(pairs[lo:hi])):
                        errors.append("mrca mismatch tree %d n %d layout %d at %d" % (k, n, layout, lo))
                if rng.random() < 0.15:
                    # the generated sources: a slice of the triangle and of a symmetric grid
                    ids = rng.choice(n_nodes, size=int(rng.integers(2, 900)), replace=False).astype(np.int64)
                    mm = len(ids)
                    total = mm * (mm - 1) // 2
                    k0 = int(rng.integers(0, total))
                    cnt = int(rng.integers(1, total - k0 + 1))
                    td, _ = dev.triangle_host(ids, k_begin=k0, k_count=cnt)
                    ii, jj = np.tril_indices(mm, -1)
                    tp = np.stack([ids[jj[k0:k0 + cnt]], ids[ii[k0:k0 + cnt]]], 1)
                    if not np.array_equal(td.view(np.int64), O.distances(tp).view(np.int64)):
                        errors.append("triangle mismatch tree %d m %d k0 %d cnt %d" % (k, mm, k0, cnt))
                    gd, _ = dev.grid_host(ids, ids, symmetric=True)
                    r, c = np.divmod(np.arange(mm * mm), mm)
                    gp = np.stack([ids[np.minimum(r, c)], ids[np.maximum(r, c)]], 1)
                    if not np.array_equal(gd.view(np.int64), O.distances(gp).view(np.int64)):
                        errors.append("grid mismatch tree %d m %d" % (k, mm))
                if rng.random() < 0.1:
                    # quartet topologies (six MRCA ids each), one chunk to several
                    nq = int(10 ** rng.uniform(0, 6.2))
                    quartets = rng.integers(0, n_nodes, (nq, 4))
                    got = dev.quartets_host(quartets)
                    lo = int(rng.integers(0, max(1, nq - 3000)))
                    if not np.array_equal(got[lo:lo + 3000], O.quartets(quartets[lo:lo + 3000])):
                        errors.append("quartet mismatch tree %d n %d at %d" % (k, nq, lo))
                if n > 10 and rng.random() < 0.1:
                    bad = pairs.copy()
                    bad[int(rng.integers(0, n)), int(rng.integers(0, 2))] = n_nodes + 3
                    try:
                        dev.distances_host(bad, True, False)
                        errors.append("missing bounds error")
                    except _capi.InvalidNodeError as e:
                        if e.node_id != n_nodes + 3:
                            errors.append("wrong bad id %r" % e.node_id)
                counts[tid] += 1
            except Exception as e:   # noqa: BLE001
                errors.append("exception %r" % (e,))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(args.threads)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    print("calls per thread:", counts, "errors:", errors[:5])
    for d in devs:
        d.close()
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
