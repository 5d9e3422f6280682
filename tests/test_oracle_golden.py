"""The oracle against every known answer the reference holds for this path.

The reference extension cannot be imported here (dendropy is absent and no
stand-in is written), so the oracle is pinned by:
  * SuchTree/tests/test.matrix -- all 225 rows (the reference's own tests use
    rel=1e-3: tests/test_SuchTree.py:56-88; here the bar is the 5 printed digits)
  * the distances and leaf ids printed in docs/examples/SuchTree_examples.md
and cross-checked against an independent pure-Python restatement of the same
reference lines.
"""
import json

import numpy as np
import pytest

from conftest import assert_bits_equal, golden_path
from oracle.oracle import OracleTree, py_distances, py_mrca, linked_pairs
from suchtree_amd.newick import flat_tree_from_newick


def _matrix(flat):
    names, want = [], []
    for line in open(golden_path("test.matrix")):
        a, b, d = line.split()
        names.append((a, b))
        want.append(float(d))
    ids = np.array([(flat.leaves[a], flat.leaves[b]) for a, b in names], dtype=np.int64)
    return names, ids, np.array(want)


def test_test_matrix_all_rows(gopher_flat):
    O = OracleTree(gopher_flat.parent, gopher_flat.distance)
    names, ids, want = _matrix(gopher_flat)
    assert len(want) == 225
    got = O.distances(ids)
    # the file carries 5 significant digits of a float32 sum
    assert np.all(np.abs(got - want) <= 5.1e-6 * np.maximum(want, 1e-3) + 1e-7)
    # scalar form (MuchTree.pyx:981-997) agrees with the bulk form bit for bit
    for (a, b), g in zip(ids[:40], got[:40]):
        assert O.distance(int(a), int(b)) == g


def test_docs_known_answers():
    known = json.load(open(golden_path("known_answers.json")))
    flat = flat_tree_from_newick(open(golden_path("host.tree")).read())
    assert flat.leaves == known["host_tree_leaves"]["value"]
    O = OracleTree(flat.parent, flat.distance)
    for row in known["host_tree_distances"]:
        a = flat.leaves[row["a"]] if isinstance(row["a"], str) else row["a"]
        b = flat.leaves[row["b"]] if isinstance(row["b"], str) else row["b"]
        assert "%f" % O.distance(a, b) == row["printed_6dp"]


def test_depth_matches_reference_definition(gopher_flat):
    O = OracleTree(gopher_flat.parent, gopher_flat.distance)
    assert O.depth == gopher_flat.depth == 9


def test_c_equals_pure_python_restatement(gopher_flat):
    O = OracleTree(gopher_flat.parent, gopher_flat.distance)
    n = gopher_flat.size
    pairs = np.array([(a, b) for a in range(n) for b in range(n)], dtype=np.int64)
    assert_bits_equal(O.distances(pairs), py_distances(gopher_flat.parent, gopher_flat.distance, pairs))
    want_m = np.array([py_mrca(gopher_flat.parent, int(a), int(b)) for a, b in pairs])
    assert np.array_equal(O.mrca_bulk(pairs), want_m)
    z = np.load(golden_path("gopher_all_pairs.npz"))   # oracle outputs captured at fixture time
    assert_bits_equal(O.distances(z["pairs"]), z["dist"])
    assert np.array_equal(O.mrca_bulk(z["pairs"]), z["mrca"])


def test_mrca_properties(gopher_flat):
    """The reference's MRCA tests are property tests (tests/test_SuchTree.py:163-183,
    tests/test_new_api.py:454-468): symmetric, an ancestor-or-self of both, and the
    deepest such node (which makes the id unique)."""
    O = OracleTree(gopher_flat.parent, gopher_flat.distance)
    parent = gopher_flat.parent

    def lineage(x):
        out = [x]
        while parent[x] != -1:
            x = int(parent[x])
            out.append(x)
        return out

    for a in range(gopher_flat.size):
        for b in range(gopher_flat.size):
            m = O.mrca(a, b)
            assert m == O.mrca(b, a)
            la, lb = lineage(a), lineage(b)
            assert m in la and m in lb
            common = [x for x in la if x in lb]
            assert common[0] == m


def test_order_of_summation_is_observable():
    """d(a,b) and d(b,a) may differ in the last ulp: the a-side is summed first."""
    from suchtree_amd import synth
    parent, dist = synth.random_binary_tree(400, seed=5)
    O = OracleTree(parent, dist)
    rng = np.random.default_rng(0)
    pairs = rng.integers(0, len(parent), (4000, 2))
    d_ab = O.distances(pairs)
    d_ba = O.distances(pairs[:, ::-1])
    assert np.allclose(d_ab, d_ba, rtol=1e-5)
    assert (d_ab != d_ba).any()


def test_strided_views_and_threads(ml_arrays):
    parent, dist, leaf_ids = ml_arrays
    O = OracleTree(parent, dist)
    rng = np.random.default_rng(1)
    pairs = rng.choice(leaf_ids, size=(3000, 2))
    want = O.distances(pairs)
    assert_bits_equal(O.distances(np.asfortranarray(pairs)), want)
    wide = np.zeros((3000, 4), dtype=np.int64)
    wide[:, ::2] = pairs
    assert_bits_equal(O.distances(wide[:, ::2]), want)
    assert_bits_equal(O.distances_mt(pairs, 3), want)


def test_digests_of_big_trees(ml_arrays, nj_arrays):
    """The oracle built on this machine reproduces the outputs captured when the
    fixtures were made (guards against a miscompiled oracle on another box)."""
    import hashlib
    dig = json.load(open(golden_path("oracle_digests.json")))
    for name, (parent, dist, leaf_ids) in (("ml", ml_arrays), ("nj", nj_arrays)):
        h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
        assert h(parent) == dig[name]["parent_sha256"]
        assert h(dist) == dig[name]["distance_sha256"]
        O = OracleTree(parent, dist)
        assert O.depth == dig[name]["depth"]
        pairs = np.random.default_rng(2).choice(leaf_ids, size=(20000, 2))
        assert h(O.distances(pairs)) == dig[name]["dist_sha256"]
        assert h(O.mrca_bulk(pairs)) == dig[name]["mrca_sha256"]


def test_linked_pairs_order():
    """MuchTree.pyx:2918-2925: k = i(i-1)/2 + j, ids_a = (ll[j,1], ll[i,1]), ids_b = (ll[j,0], ll[i,0])."""
    ll = np.array([[10, 0], [11, 2], [12, 4], [13, 6]], dtype=np.int64)
    a, b = linked_pairs(ll)
    assert a.tolist() == [[0, 2], [0, 4], [2, 4], [0, 6], [2, 6], [4, 6]]
    assert b.tolist() == [[10, 11], [10, 12], [11, 12], [10, 13], [11, 13], [12, 13]]


def test_quartet_topologies_restatement():
    """MuchTree.pyx:1331-1376: the sister pair is the one whose MRCA id is unique among the six."""
    from itertools import permutations
    t = flat_tree_from_newick("(((A,B),C),(D,E));")
    O = OracleTree(t.parent, t.distance)
    L = t.leaves
    for perm in permutations("ABCD"):
        row = O.quartets(np.array([[L[x] for x in perm]]))[0].tolist()
        assert {frozenset(row[:2]), frozenset(row[2:])} == {frozenset((L["A"], L["B"])), frozenset((L["C"], L["D"]))}
    row = O.quartets(np.array([[L["D"], L["A"], L["E"], L["C"]]]))[0].tolist()
    assert {frozenset(row[:2]), frozenset(row[2:])} == {frozenset((L["D"], L["E"])), frozenset((L["A"], L["C"]))}
    # the output row is a permutation of the input row, by the table at :1319-1320
    q = np.array([[L["B"], L["D"], L["A"], L["E"]]])
    assert O.quartets(q)[0].tolist() == [L["B"], L["A"], L["D"], L["E"]]


def test_spectral_properties_csv_gopher_lice():
    """A reference-held number for the two-tree graph Laplacian that does not depend on node numbering:
    data/spectral_properties.csv of the reference (written in 2017 by docs/old_notebooks/example_3.ipynb with real
    dendropy) holds, for "Gopher, Lice" at additions = deletions = swaps = 0, skew and kurtosis of a Gaussian
    KDE of the Laplacian's spectrum on linspace(-0.5, 1.5, 100) and the gap between its two largest eigenvalues.
    That code base counted the resolved-polytomy edges (length epsilon) in the mean edge length that weights
    the links; today's reference masks them (MuchTree.pyx:3120-3121), and so does the oracle.  With that one
    difference applied, the oracle's adjacency (trees from this repo's Newick ingest, gopher-louse fixtures)
    reproduces all three numbers to the 12 digits the file prints (scripts/spectral_pins.py tries every study)."""
    import pandas as pd
    from scipy.stats import gaussian_kde, kurtosis, skew
    from oracle import oracle as orc
    from suchtree_amd import SuchTree
    from suchtree_amd.linked import SuchLinkedTrees
    d = golden_path("gopher_louse")
    A, B = SuchTree(d + "/gopher.tree"), SuchTree(d + "/lice.tree")
    SLT = SuchLinkedTrees(A, B, pd.read_csv(d + "/links.csv", index_col=0))
    assert (A.num_leaves, B.num_leaves, SLT.n_links) == (15, 17, 17)          # the row's n_hosts, n_guests, n_links
    fa, fb = A._flat, B._flat
    aj = orc.linked_adjacency((fa.parent, fa.left, fa.right, fa.distance), (fb.parent, fb.left, fb.right, fb.distance),
                              SLT.linklist, SLT.subset_a_root, SLT.subset_b_root, A.polytomy_epsilon, B.polytomy_epsilon)
    na = A.size
    ta = orc.tree_adjacency(fa.parent, fa.left, fa.right, fa.distance, A.root_node, A.polytomy_epsilon)[0]
    tb = orc.tree_adjacency(fb.parent, fb.left, fb.right, fb.distance, B.root_node, B.polytomy_epsilon)[0]
    is_link = np.zeros_like(aj, dtype=bool)
    is_link[:na, na:] = aj[:na, na:] > 0
    is_link[na:, :na] = aj[na:, :na] > 0
    assert is_link.sum() == 2 * 17
    aj[is_link] = (ta[ta > 0].mean() / ta.max() + tb[tb > 0].mean() / tb.max()) / 2.0      # the 2017 link weight
    lam = np.linalg.eigvalsh(orc.linked_laplacian(aj))
    sd = gaussian_kde(lam).pdf(np.linspace(-0.5, 1.5, 100))
    assert abs((lam[-1] - lam[-2]) - 0.351291850306) < 1e-11
    assert abs(skew(sd) - (-0.285380791319)) < 1e-11
    assert abs(kurtosis(sd) - (-0.801738279906)) < 1e-11



def test_docs_correlations_between_ml_and_nj_tree(ml_arrays, nj_arrays):
    """A reference-held number for config 2's trees: docs/examples/SuchTree_examples.md:296-352 draws one
    million random pairs of taxon names, takes their distances in data/bigtrees/ml.tree and nj.tree with
    distances_by_name and prints Spearman's rs 0.961, Kendall's tau 0.824, Pearson's r 0.969 (real dendropy,
    unseeded pairs, three decimals).  Here: the flat-array fixtures of both trees (this repo's parser), the
    committed taxon map between them, 300,000 seeded pairs through the oracle.  Sampling noise at that size is
    below 1e-3; the bar is the printed precision."""
    import os
    from scipy.stats import kendalltau, pearsonr, spearmanr
    p1, d1, leaves1 = ml_arrays
    p2, d2, _ = nj_arrays
    nj_of = np.load(golden_path("ml_nj_leaf_map.npz"))["nj_id_of_ml_leaf"].astype(np.int64)
    assert len(nj_of) == len(leaves1) == 54327 and len(set(nj_of.tolist())) == 54327
    idx = np.random.default_rng(11).integers(0, len(leaves1), (300_000, 2))
    cores = len(os.sched_getaffinity(0))
    D1 = OracleTree(p1, d1).distances_mt(leaves1[idx], cores)
    D2 = OracleTree(p2, d2).distances_mt(nj_of[idx], cores)
    assert abs(spearmanr(D1, D2)[0] - 0.961) < 0.0015
    assert abs(kendalltau(D1, D2)[0] - 0.824) < 0.0015
    assert abs(pearsonr(D1, D2)[0] - 0.969) < 0.0015


@pytest.mark.parametrize("which", ["gopher", "ml", "nj"])
def test_oracle_reproduces_the_reference_s_compiled_hot_path_vectors(which, gopher_flat, ml_arrays, nj_arrays):
    """tests/golden/ref_hotpath_vectors.npz: seeded inputs and the outputs of the reference's OWN compiled _distances / _mrca /
    _quartet_topologies (scripts/make_ref_golden.py, run where /root/reference and oracle/_ref/libref_hotpath.so exist) -- data, so
    the pin at full precision (last-ulp summation order, every MRCA id) holds wherever the library itself is not available."""
    from oracle.oracle import OracleTree
    z = np.load(golden_path("ref_hotpath_vectors.npz"))
    parent, dist = {"gopher": (gopher_flat.parent, gopher_flat.distance), "ml": ml_arrays[:2], "nj": nj_arrays[:2]}[which]
    O = OracleTree(parent, dist)
    assert O.depth == int(z[which + "_depth"])
    pairs = z[which + "_pairs"].astype(np.int64)
    assert np.array_equal(O.distances(pairs).view(np.int64), z[which + "_dist"].astype(np.float64).view(np.int64))
    assert np.array_equal(O.mrca_bulk(pairs), z[which + "_mrca"])
    assert np.array_equal(O.quartets(z[which + "_quartets"].astype(np.int64)), z[which + "_topologies"].astype(np.int64))
