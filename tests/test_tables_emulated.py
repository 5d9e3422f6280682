"""Product table construction + per-pair functions, emulated on the host,
against the oracle.  Covers the host logic of both kernel families without a
GPU (the kernels themselves are checked by the -m gpu tests)."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle.oracle import OracleTree
from suchtree_amd import synth


def _check(emulator, parent, dist, pairs, expect_canopy=None):
    O = OracleTree(parent, dist)
    want_d, want_m = O.distances(pairs), O.mrca_bulk(pairs)
    infos = {}
    for strategy in ("walk", "canopy"):
        d, m, info = emulator.run(parent, dist, pairs, strategy)
        assert_bits_equal(d, want_d, "%s distances" % strategy)
        assert np.array_equal(m, want_m), "%s mrca" % strategy
        assert info.depth == O.depth
        infos[strategy] = info
    _, m_only, _ = emulator.run(parent, dist, pairs, "walk", want_dist=False)
    assert np.array_equal(m_only, want_m)
    if expect_canopy is not None:
        i = infos["canopy"]
        assert (i.canopy_nodes, i.understory_max, i.record_bytes) == expect_canopy
    return infos["canopy"]


def test_gopher_every_pair(emulator, gopher_flat):
    n = gopher_flat.size
    pairs = np.array([(a, b) for a in range(n) for b in range(n)], dtype=np.int64)
    _check(emulator, gopher_flat.parent, gopher_flat.distance, pairs)


def test_ml_tree_leaf_internal_and_near_pairs(emulator, ml_arrays):
    parent, dist, leaf_ids = ml_arrays
    rng = np.random.default_rng(0)
    n = len(parent)
    info = _check(emulator, parent, dist, rng.choice(leaf_ids, size=(20000, 2)))
    assert info.parity == 1 and info.canopy_nodes <= 16384 and info.record_bytes == 64
    _check(emulator, parent, dist, rng.integers(0, n, (20000, 2)))
    a = np.arange(0, 30000)
    near = np.stack([a, a + rng.integers(0, 7, a.size)], 1)     # same-portal and understory MRCAs
    _check(emulator, parent, dist, near)


@pytest.mark.parametrize("levels,expect", [(3, None), (10, None), (14, (16383, 1, 16)), (17, (2047, 7, 64))])
def test_balanced_trees(emulator, levels, expect):
    parent, dist = synth.balanced_tree(levels)
    n = len(parent)
    rng = np.random.default_rng(levels)
    _check(emulator, parent, dist, rng.integers(0, n, (30000, 2)), expect)
    a = np.arange(0, min(n - 9, 30000))
    _check(emulator, parent, dist, np.stack([a, a + rng.integers(0, 9, a.size)], 1))


def test_balanced_2_20_geometry(emulator):
    """BASELINE's headline tree: 16383-node canopy (top 14 levels, 128 KiB of LDS),
    7-node understory chains, 64-byte records."""
    parent, dist = synth.balanced_tree(20)
    pairs = synth.random_leaf_pairs(1 << 20, 20000, seed=3)
    info = _check(emulator, parent, dist, pairs, (16383, 7, 64))
    assert info.parity == 1 and info.n_leaves == 1 << 20


def test_deep_and_random_trees(emulator):
    rng = np.random.default_rng(9)
    parent, dist = synth.caterpillar_tree(3000)
    _check(emulator, parent, dist, rng.integers(0, len(parent), (5000, 2)))
    parent, dist = synth.random_binary_tree(50000, seed=1, zero_fraction=0.1)
    _check(emulator, parent, dist, rng.integers(0, len(parent), (40000, 2)))


def test_tiny_trees(emulator):
    parent, dist = synth.random_binary_tree(1, seed=1)
    _check(emulator, parent, dist, np.array([[0, 0]]))
    parent, dist = synth.random_binary_tree(2, seed=1)
    _check(emulator, parent, dist, np.array([[0, 0], [0, 2], [2, 0], [1, 2], [0, 1], [1, 1]]))


def test_special_float_values(emulator):
    """Denormals, signed zeros, huge and negative lengths must add exactly as on the CPU."""
    parent, dist = synth.random_binary_tree(3000, seed=4)
    rng = np.random.default_rng(4)
    dist = dist.copy()
    k = rng.integers(0, len(dist), 600)
    dist[k[:100]] = np.float32(1e-42)       # denormal
    dist[k[100:200]] = np.float32(-0.0)
    dist[k[200:300]] = np.float32(3e38)
    dist[k[300:400]] = np.float32(-1.5)
    dist[k[400:500]] = np.float32(2.220446e-16)
    dist[k[500:]] = np.float32(1.17549435e-38)
    _check(emulator, parent, dist, rng.integers(0, len(parent), (30000, 2)))


def test_caterpillar_beyond_canopy_is_refused(emulator):
    """More than 16384 backbone nodes with long chains: the canopy family must say no."""
    parent, dist = synth.caterpillar_tree(40000)
    with pytest.raises(RuntimeError, match="canopy not admitted"):
        emulator.run(parent, dist, np.array([[0, 2]]), "canopy")
    d, m, _ = emulator.run(parent, dist, np.array([[0, 79998]]), "walk")
    O = OracleTree(parent, dist)
    assert d[0] == O.distances(np.array([[0, 79998]]))[0]


def test_bad_trees_are_rejected(emulator):
    with pytest.raises(RuntimeError):
        emulator.run(np.array([1, 2, 0]), np.zeros(3), np.array([[0, 1]]), "walk")   # cycle
    with pytest.raises(RuntimeError):
        emulator.run(np.array([-1, -1]), np.zeros(2), np.array([[0, 1]]), "walk")    # two roots


def _general_tree(rng, n, max_children):
    """Random rooted tree with arbitrary arity and arbitrary node numbering (what a C-ABI caller
    other than the facade may hand over): parent[] only, no in-order structure."""
    order = rng.permutation(n)
    parent = np.full(n, -1, dtype=np.int32)
    n_child = np.zeros(n, dtype=np.int64)
    for k in range(1, n):
        while True:
            p = order[rng.integers(max(0, k - 50), k)]
            if n_child[p] < max_children:
                break
        parent[order[k]] = p
        n_child[p] += 1
    dist = rng.uniform(0.01, 2.0, n).astype(np.float32)
    dist[order[0]] = -1.0
    return parent, dist


@pytest.mark.parametrize("n,max_children", [(50, 1), (400, 2), (3000, 3), (20000, 8)])
def test_general_trees_through_the_c_tables(emulator, n, max_children):
    """The table builder does not rely on the tree being binary or numbered in order: unary
    chains, wide polytomies and shuffled ids go through both families (records in identity
    order, `parity` off) and agree with the plain-Python restatement of the reference."""
    from oracle.oracle import py_distances, py_mrca
    rng = np.random.default_rng(n)
    parent, dist = _general_tree(rng, n, max_children)
    pairs = rng.integers(0, n, (3000, 2))
    want_d = py_distances(parent, dist, pairs)
    want_m = np.array([py_mrca(parent, int(a), int(b)) for a, b in pairs])
    for strategy in ("walk", "canopy"):
        d, m, info = emulator.run(parent, dist, pairs, strategy)
        assert_bits_equal(d, want_d, strategy)
        assert np.array_equal(m, want_m), strategy
    assert info.parity == 0


def test_config4_complete_tree(emulator):
    """BASELINE config 4's tree as SURVEY 8d words it: complete binary tree, last level partially
    filled, 100,000 leaves, seed 44.  Shape checks, then both table families against the oracle."""
    parent, dist = synth.complete_tree(100_000, seed=44)
    n = len(parent)
    assert n == 199_999 and int((parent < 0).sum()) == 1
    depth = np.zeros(n, dtype=np.int64)
    from suchtree_amd.newick import node_depths
    depth = node_depths(parent)
    leaf_depth = depth[0::2]                                   # leaves are the even ids
    assert set(np.unique(leaf_depth).tolist()) == {16, 17}     # 2^16 < 100000 <= 2^17
    assert np.all(np.diff(leaf_depth) <= 0)                    # the deep (last-level) leaves are the leftmost
    assert int((leaf_depth == 17).sum()) == 2 * (100_000 - 65_536)
    small_p, small_d = synth.complete_tree(11, seed=44)
    from suchtree_amd.newick import flat_tree_from_newick
    assert np.array_equal(flat_tree_from_newick(synth.to_newick(small_p, small_d)).parent, small_p)
    rng = np.random.default_rng(44)
    leaf_pairs = rng.integers(0, 100_000, (30_000, 2)) * 2
    info = _check(emulator, parent, dist, leaf_pairs)
    assert info.parity == 1 and info.n_leaves == 100_000
    _check(emulator, parent, dist, rng.integers(0, n, (20_000, 2)))
