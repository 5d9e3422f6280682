"""The reference's own published benchmark (docs/benchmarks.md:31-63): 1,000,000 random leaf-name
pairs through distances_by_name on each of the two 54,327-leaf trees (ml.tree, nj.tree).
Published: 10.1 s for the two calls on an i7-3770S (one thread)."""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleTree   # noqa: E402  (checker only)
from suchtree_amd import SuchTree      # noqa: E402


def main():
    out = {"benchmark": "2 x 1,000,000 distances_by_name (ml.tree + nj.tree)", "published_seconds_i7_3770S": 10.1}
    total = 0.0
    for name in ("ml", "nj"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "%s_tree.npz" % name))
        leaf_ids = z["leaf_ids"]
        names = ["taxon_%d" % i for i in range(len(leaf_ids))]        # names are not shipped; ids are
        T = SuchTree((z["parent"], z["distance"], names)).to_device()
        rng = random.Random(5)
        pairs = [(rng.choice(names), rng.choice(names)) for _ in range(1_000_000)]
        T.distances_by_name(pairs[:1000])
        t0 = time.perf_counter()
        d = T.distances_by_name(pairs)
        dt = time.perf_counter() - t0
        total += dt
        O = OracleTree(z["parent"], z["distance"])
        ids = np.array([(T.leaves[a], T.leaves[b]) for a, b in pairs[:100000]])
        ok = np.array_equal(np.array(d[:100000]).view(np.int64), O.distances(ids).view(np.int64))
        out[name] = {"seconds": dt, "pairs_per_s": 1e6 / dt, "parity": "bit-exact on 100000" if ok else "MISMATCH"}
    out["total_seconds"] = total
    print(json.dumps(out))


if __name__ == "__main__":
    main()
