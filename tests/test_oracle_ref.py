"""The oracle's restatement against the REFERENCE's own compiled hot path (oracle/_ref/libref_hotpath.so: the C that Cython
generated from SuchTree._distances / _mrca, shipped in the reference's repository, compiled where it lies -- oracle/ref_harness.c).
Distances bit for bit, MRCA ids exactly, on the reference's own fixture, its big trees, synthetic shapes, special float values and
strided id arrays.  This is what pins the last-ulp summation order (a side first, float32) and every MRCA id by a reference RUN,
where the reference's printed goldens only reach 5 significant digits.  Skipped when the library is not there (no /root/reference
and no prebuilt file)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as orc
from suchtree_amd import synth

pytestmark = pytest.mark.skipif(orc.ref_lib() is None, reason="oracle/_ref/libref_hotpath.so not available")


def _both(parent, dist):
    O = orc.OracleTree(parent, dist)
    return O, orc.RefTree(parent, dist, depth=O.depth)


def _same(O, R, pairs, what):
    d_o, d_r = O.distances(pairs), R.distances(pairs)
    assert np.array_equal(d_o.view(np.int64), d_r.view(np.int64)), what
    assert np.array_equal(O.mrca_bulk(pairs), R.mrca_bulk(pairs)), what
    return d_r


def _mixed_pairs(rng, parent, n):
    """uniform node pairs (leaves and internal nodes), near pairs, (x, x), a node with one of its ancestors, both orders"""
    size = len(parent)
    root = int(np.flatnonzero(parent < 0)[0])
    a = rng.integers(0, size, n)
    up = a.copy()
    for _ in range(int(rng.integers(1, 40))):
        up = np.where(parent[up] >= 0, parent[up], root)
    near = np.clip(a + rng.integers(-20, 21, n), 0, size - 1)
    return np.concatenate([rng.integers(0, size, (n, 2)), np.stack([a, near], 1), np.stack([a, a], 1), np.stack([a, up], 1),
                           np.stack([up, a], 1), np.stack([np.full(64, root), a[:64]], 1)]).astype(np.int64)


def test_reference_code_reproduces_its_own_test_matrix(gopher_flat):
    """SuchTree/tests/test.matrix (225 name pairs, 5 significant digits) through the reference's compiled _distances."""
    R = orc.RefTree(gopher_flat.parent, gopher_flat.distance, depth=gopher_flat.depth)
    rows = [l.split() for l in open(os.path.join(GOLDEN, "test.matrix")) if l.strip()]
    pairs = np.array([[gopher_flat.leaves[a], gopher_flat.leaves[b]] for a, b, _ in rows], dtype=np.int64)
    want = np.array([float(v) for _, _, v in rows])
    got = R.distances(pairs)
    assert np.allclose(got, want, rtol=5e-5, atol=1e-9)
    assert R.depth == 9      # (gopher 9)


def test_gopher_every_node_pair(gopher_flat):
    O, R = _both(gopher_flat.parent, gopher_flat.distance)
    n = len(gopher_flat.parent)
    pairs = np.array([[a, b] for a in range(n) for b in range(n)], dtype=np.int64)
    _same(O, R, pairs, "gopher, all 29 x 29 node pairs")


@pytest.mark.parametrize("which", ["ml", "nj"])
def test_bigtrees(which, ml_arrays, nj_arrays):
    parent, dist, leaf_ids = ml_arrays if which == "ml" else nj_arrays
    O, R = _both(parent, dist)
    rng = np.random.default_rng(17)
    _same(O, R, _mixed_pairs(rng, parent, 40_000), which + " mixed")
    _same(O, R, rng.choice(leaf_ids.astype(np.int64), size=(100_000, 2)), which + " leaf pairs")
    # several threads on contiguous chunks give what one thread gives
    p = rng.choice(leaf_ids.astype(np.int64), size=(50_000, 2))
    assert np.array_equal(R.distances(p, 5).view(np.int64), R.distances(p).view(np.int64))


def test_synthetic_shapes_and_special_lengths():
    rng = np.random.default_rng(23)
    trees = [synth.balanced_tree(12), synth.caterpillar_tree(700), synth.random_binary_tree(5000, seed=4, zero_fraction=0.2),
             synth.skewed_tree(rng, 20_000, 0.9), synth.complete_tree(1000, seed=44)]
    for k, (parent, dist) in enumerate(trees):
        dist = dist.copy()
        if k % 2 == 0:      # zeros, negatives (NJ trees), denormals, huge values, the reference's epsilon
            idx = rng.integers(0, len(dist), max(1, len(dist) // 8))
            dist[idx] = rng.choice(np.array([0.0, -0.25, 1e-42, 3e37, 2.220446e-16, -0.0], np.float32), len(idx))
            dist[parent < 0] = -1.0
        O, R = _both(parent, dist)
        _same(O, R, _mixed_pairs(rng, parent, 6000), "tree %d" % k)


def test_strided_id_arrays(ml_arrays):
    """The reference takes any `long[:, :]` memoryview (MuchTree.pyx:913): Fortran order, sliced rows, swapped columns."""
    parent, dist, _ = ml_arrays
    O, R = _both(parent, dist)
    rng = np.random.default_rng(5)
    base = rng.integers(0, len(parent), (4000, 4)).astype(np.int64)
    for view in (np.asfortranarray(base[:, :2]), base[::3, 1:3], base[:, ::2], base[:, 2::-2][:, ::-1]):
        if view.strides[0] < 0 or view.strides[1] < 0:
            continue
        want = O.distances(np.ascontiguousarray(view))
        assert np.array_equal(R.distances(view).view(np.int64), want.view(np.int64))
        assert np.array_equal(R.mrca_bulk(view), O.mrca_bulk(np.ascontiguousarray(view)))


def test_quartet_topologies(ml_arrays, gopher_flat):
    """SuchTree._quartet_topologies (pyx:1331-1376: six _mrca calls and the pick of the unique one) -- the reference's compiled code
    against the restatement, leaves and internal nodes, repeated members."""
    rng = np.random.default_rng(41)
    for parent, dist in ((ml_arrays[0], ml_arrays[1]), (gopher_flat.parent, gopher_flat.distance), synth.balanced_tree(10)):
        O, R = _both(parent, dist)
        q = rng.integers(0, len(parent), (20_000, 4)).astype(np.int64)
        q[::50, 1] = q[::50, 0]
        assert np.array_equal(O.quartets(q), R.quartets(q))
