"""Golden vectors of the hot path from the REFERENCE's own compiled code (this container only: needs oracle/_ref/libref_hotpath.so, i.e.
/root/reference): seeded inputs and the outputs of the reference's SuchTree._distances / _mrca / _quartet_topologies, stored as data
under tests/golden/ref_hotpath_vectors.npz so that the oracle (and the GPU path) stay pinned at full precision -- last-ulp summation
order, every MRCA id -- wherever the library itself is not available.
    python scripts/make_ref_golden.py
Distances are stored as float32 (they ARE float32 values: the reference accumulates in C float, MuchTree.c:32341), ids as int32."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc      # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def mixed_pairs(rng, parent, n):
    size = len(parent)
    root = int(np.flatnonzero(parent < 0)[0])
    a = rng.integers(0, size, n)
    up = a.copy()
    for _ in range(int(rng.integers(1, 40))):
        up = np.where(parent[up] >= 0, parent[up], root)
    near = np.clip(a + rng.integers(-20, 21, n), 0, size - 1)
    return np.concatenate([rng.integers(0, size, (2 * n, 2)), np.stack([a, near], 1), np.stack([a[: n // 4], a[: n // 4]], 1),
                           np.stack([a, up], 1), np.stack([up, a], 1)]).astype(np.int64)


def main():
    if orc.ref_lib() is None:
        raise SystemExit("oracle/_ref/libref_hotpath.so is not available: run `make -C oracle ref` where /root/reference exists")
    out = {}
    from suchtree_amd import SuchTree
    flat = SuchTree(os.path.join(GOLDEN, "test.tree"))._flat      # (host-side ingest only: no GPU is touched before a query)
    trees = {"gopher": (np.asarray(flat.parent, np.int32), np.asarray(flat.distance, np.float32))}
    for name in ("ml", "nj"):
        z = np.load(os.path.join(GOLDEN, "%s_tree.npz" % name))
        trees[name] = (z["parent"].astype(np.int32), z["distance"].astype(np.float32))
    rng = np.random.default_rng(20261004)
    for name, (parent, dist) in trees.items():
        R = orc.RefTree(parent, dist)
        n = len(parent)
        pairs = (np.array([[a, b] for a in range(n) for b in range(n)], dtype=np.int64) if name == "gopher" else mixed_pairs(rng, parent, 2000))
        d = R.distances(pairs)
        assert np.array_equal(d, d.astype(np.float32).astype(np.float64))      # float32 values, widened
        q = rng.integers(0, n, (1500, 4)).astype(np.int64)
        out["%s_depth" % name] = np.int32(R.depth)
        out["%s_pairs" % name] = pairs.astype(np.int32)
        out["%s_dist" % name] = d.astype(np.float32)
        out["%s_mrca" % name] = R.mrca_bulk(pairs)
        out["%s_quartets" % name] = q.astype(np.int32)
        out["%s_topologies" % name] = R.quartets(q).astype(np.int32)
        print(name, "depth", R.depth, len(pairs), "pairs", len(q), "quartets")
    path = os.path.join(GOLDEN, "ref_hotpath_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
