"""CPU only: the oracle's restatement against the reference's own compiled hot path (oracle/_ref/libref_hotpath.so) on random trees --
random shapes, sizes and numberings, special branch lengths, random node pairs and quartets -- distances bit for bit, MRCA ids and quartet
topologies exactly.    python scripts/fuzz_oracle_vs_ref.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc      # noqa: E402
from suchtree_amd import synth      # noqa: E402
from test_tables_emulated import _general_tree      # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    if orc.ref_lib() is None:
        raise SystemExit("oracle/_ref/libref_hotpath.so is not available")
    rng = np.random.default_rng(seed)
    print("seed", seed, flush=True)
    t_end = time.time() + budget
    trees = pairs_total = quartets_total = 0
    while time.time() < t_end:
        kind = int(rng.integers(0, 6))
        if kind == 0:
            parent, dist = synth.balanced_tree(int(rng.integers(1, 15)))
        elif kind == 1:
            parent, dist = _general_tree(rng, int(rng.integers(2, 20000)), int(rng.integers(1, 9)))      # any arity, any numbering
        elif kind == 2:
            parent, dist = synth.caterpillar_tree(int(rng.integers(2, 3000)))
        else:
            parent, dist = synth.skewed_tree(rng, int(2 ** rng.uniform(1, 16)), float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.97, 0.995])))
        n = len(parent)
        if kind == 5:      # ids that are not in-order positions
            new_id = rng.permutation(n)
            p2 = np.empty(n, np.int32)
            d2 = np.empty(n, np.float32)
            p2[new_id] = np.where(parent >= 0, new_id[np.maximum(parent, 0)], -1)
            d2[new_id] = dist
            parent, dist = p2, d2
        if rng.integers(0, 3) == 0:
            dist = np.array(dist, np.float32)
            k = rng.integers(0, n, max(1, n // 6))
            dist[k] = rng.choice(np.array([0.0, -0.5, 1e-30, 1e-44, 3e37, 2.220446e-16, -0.0, 1.0], np.float32), len(k))
            dist[np.asarray(parent) < 0] = -1.0
        O = orc.OracleTree(parent, dist)
        R = orc.RefTree(parent, dist, depth=O.depth)
        m = int(rng.integers(1, 20000))
        pairs = rng.integers(0, n, (m, 2)).astype(np.int64)
        q = rng.integers(0, n, (int(rng.integers(1, 3000)), 4)).astype(np.int64)
        ok = (np.array_equal(O.distances(pairs).view(np.int64), R.distances(pairs).view(np.int64)) and np.array_equal(O.mrca_bulk(pairs), R.mrca_bulk(pairs))
              and np.array_equal(O.quartets(q), R.quartets(q)))
        if not ok:
            print("MISMATCH: kind %d n %d seed %d tree %d" % (kind, n, seed, trees), flush=True)
            return 1
        trees += 1
        pairs_total += m
        quartets_total += len(q)
    print("oracle == reference's compiled code on %d trees, %d pairs, %d quartets" % (trees, pairs_total, quartets_total))
    return 0


if __name__ == "__main__":
    sys.exit(main())
