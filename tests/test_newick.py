"""Newick ingest: node ids, epsilon rule, support, errors.

Pinned against dendropy-produced outputs printed in the reference's docs and
against the structural facts the reference's tests assert
(tests/test_SuchTree.py:35-44 children ids, :22-24 polytomy resolution).
"""
import json
import os

import numpy as np
import pytest

from conftest import golden_path
from suchtree_amd import synth
from suchtree_amd.exceptions import TreeStructureError
from suchtree_amd.newick import (EPSILON, flat_tree_from_arrays, flat_tree_from_newick, node_depths)


def test_host_tree_leaf_ids_match_docs():
    known = json.load(open(golden_path("known_answers.json")))
    t = flat_tree_from_newick(open(golden_path("host.tree")).read())
    assert t.leaves == known["host_tree_leaves"]["value"]
    assert list(t.leaves.values()) == sorted(t.leaves.values())   # dict order = in-order


def test_simple_polytomy_string():
    # test_tree_str of the reference's tests: '(A,B,(C,D));'
    t = flat_tree_from_newick("(A,B,(C,D));")
    assert t.leaves == {"C": 0, "D": 2, "A": 4, "B": 6}
    assert (t.root, t.size, t.depth, t.num_leaves) == (3, 7, 3, 4)
    eps32 = np.float32(EPSILON)
    assert np.all(t.distance[[0, 1, 2, 4, 5, 6]] == eps32)     # missing lengths -> epsilon (pyx:188-189)
    assert t.distance[3] == -1.0 and t.parent[3] == -1
    assert t.left[5] == 4 and t.right[5] == 6 and t.parent[5] == 3   # new node joins the first two children


def test_gopher_tree_structure(gopher_flat):
    t = gopher_flat
    assert (t.size, t.num_leaves, t.depth, t.root) == (29, 15, 9, 25)
    # trifurcating root: (Ttal, Tbot, X) -> node 27 = (Ttal=26, Tbot=28), epsilon above it
    assert t.leaves["Ttal"] == 26 and t.leaves["Tbot"] == 28
    assert (t.left[27], t.right[27], t.parent[27]) == (26, 28, 25)
    assert t.distance[27] == np.float32(EPSILON)
    assert t.distance[26] == np.float32(0.07713)
    # children ids are consistent with in-order numbering (left < node < right)
    for i in t.internal_nodes:
        assert t.left[i] < i < t.right[i]
        assert t.parent[t.left[i]] == i and t.parent[t.right[i]] == i
    assert sorted(t.leaf_nodes) == list(range(0, 29, 2))


def test_published_sizes_of_bigtrees(ml_arrays, nj_arrays):
    known = json.load(open(golden_path("known_answers.json")))["bigtrees_sizes"]
    for parent, dist, leaf_ids in (ml_arrays, nj_arrays):
        assert len(parent) == known["nodes"] and len(leaf_ids) == known["leaves"]
        assert np.array_equal(leaf_ids, np.arange(0, len(parent), 2))


def test_four_way_polytomy_and_zero_lengths():
    t = flat_tree_from_newick("(A:1,B:2,C:0,D:4)R:9;")
    # [A,B,C,D] -> [C,D,(A,B)] -> [(A,B),(C,D)]
    assert t.leaves == {"A": 0, "B": 2, "C": 4, "D": 6}
    assert t.root == 3
    assert t.distance[4] == np.float32(EPSILON)        # explicit zero -> epsilon (pyx:191-192)
    assert t.distance[1] == np.float32(EPSILON) and t.distance[5] == np.float32(EPSILON)
    assert t.distance[3] == -1.0                        # root length ignored


def test_labels_comments_quotes_support():
    t = flat_tree_from_newick("[&R] ((a_b:1,'c d''e':2)0.95:3,(x:1,y:1)boot:2)[note];\n")
    assert list(t.leaves) == ["a_b", "c d'e", "x", "y"]
    assert t.support[1] == np.float32(0.95)
    assert t.support[5] == -1.0                          # non-numeric label
    assert all(t.support[i] == -1.0 for i in t.leaf_nodes)
    assert t.distance[t.leaves["c d'e"]] == 2.0


def test_negative_lengths_pass_through():
    t = flat_tree_from_newick("((A:-0.5,B:1):1,C:2);")
    assert t.distance[t.leaves["A"]] == np.float32(-0.5)


def test_single_leaf_and_errors():
    t = flat_tree_from_newick("A;")
    assert (t.size, t.root, t.depth) == (1, 0, 1) and t.leaves == {"A": 0}
    with pytest.raises(TypeError):
        flat_tree_from_newick("((A:1,B:1):1);")          # unifurcation: no in-order traversal
    with pytest.raises(TreeStructureError):
        flat_tree_from_newick("((A,B);")
    with pytest.raises(TreeStructureError):
        flat_tree_from_newick("(A:x,B);")
    with pytest.raises(TreeStructureError):
        flat_tree_from_newick("")


@pytest.mark.parametrize("maker", [
    lambda: synth.balanced_tree(6),
    lambda: synth.caterpillar_tree(40),
    lambda: synth.random_binary_tree(200, seed=3),
])
def test_synthetic_arrays_survive_a_newick_round_trip(maker):
    parent, dist = maker()
    text = synth.to_newick(parent, dist)
    t = flat_tree_from_newick(text)
    assert np.array_equal(t.parent, parent)
    assert np.array_equal(t.distance, dist)
    f = flat_tree_from_arrays(parent, dist)
    assert np.array_equal(f.left, t.left) and np.array_equal(f.right, t.right)
    assert f.depth == t.depth and f.root == t.root and f.leaves == t.leaves


def test_balanced_tree_shape():
    parent, dist = synth.balanced_tree(10)
    assert len(parent) == 2047 and parent[1023] == -1
    d = node_depths(parent)
    assert d.max() == 10 and np.all(d[0::2] == 10)
    assert dist.dtype == np.float32 and dist[1023] == -1.0


def test_flat_arrays_validation():
    with pytest.raises(TreeStructureError):
        flat_tree_from_arrays(np.array([-1, -1, 1]), np.zeros(3))
    with pytest.raises(TreeStructureError):
        flat_tree_from_arrays(np.array([1, 2, 1]), np.zeros(3))       # cycle, no root
    with pytest.raises(TreeStructureError):
        flat_tree_from_arrays(np.array([1, -1, 1, 1, 1]), np.zeros(5))  # not binary


# ---- native ingest (csrc/newick_parse.cpp) against the pure-Python statement ----------------

def _same_tree(a, b):
    for k in ("parent", "left", "right", "support", "distance"):
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and np.array_equal(x, y, equal_nan=True), k
    assert (a.root, a.depth, a.size) == (b.root, b.depth, b.size)
    assert list(a.leaves.items()) == list(b.leaves.items()) and a.leaf_nodes == b.leaf_nodes
    assert np.array_equal(a.internal_nodes, b.internal_nodes)


def _native_available():
    from suchtree_amd import _capi
    return _capi.newick_native("(A,B);") is not None


@pytest.mark.skipif(not _native_available(), reason="libsuchtree_hip.so not built")
class TestNativeNewick:
    def test_fixture_trees(self):
        from suchtree_amd.newick import _flat_tree_from_newick_py
        import glob
        files = [golden_path("test.tree"), golden_path("host.tree")] + \
            sorted(glob.glob(golden_path("gopher_louse") + "/*.tree")) + \
            sorted(glob.glob(golden_path("fish_worm") + "/*.tree"))
        for f in files:
            text = open(f).read()
            _same_tree(flat_tree_from_newick(text), _flat_tree_from_newick_py(text))

    def test_syntax_corners(self):
        from suchtree_amd import _capi
        from suchtree_amd.newick import _flat_tree_from_newick_py
        handled = ["(A,B,(C,D));", "(A:1,B:2,C:0,D:4)R:9;", "A;", "((A:-0.5,B:1):1,C:2);", "(A:1e-3,B:.5)\n",
                   "[&R] ((a_b:1,'c d''e':2)0.95:3,(x:1,y:1)boot:2)[note];\n", "(A:1,B:2);(C,D);",
                   "( A : 1 ,\n B : 2 ) ;", "((A,B)1e2,(C,D)-3.5)root;", "(A[x]:1[y],B:2);", "('':1,B:2);",
                   "(A:0.0,B:-0.0,(C:5E-1,D:+2)n9);", "(((((A,B),C),D),E),F,G,H,I);"]
        for text in handled:
            assert _capi.newick_native(text) is not None, text
            _same_tree(flat_tree_from_newick(text), _flat_tree_from_newick_py(text))
        # tokens whose Python meaning the native parser does not claim: it declines, Python decides
        declined = ["((A,B)n1,(C,D)'1.5 ')1_0;", "((A,B)inf,(C,D)NaN);", "(A:1_0,B:2);", "(é:1,B:2);"]
        for text in declined:
            assert _capi.newick_native(text) is None, text
            _same_tree(flat_tree_from_newick(text), _flat_tree_from_newick_py(text))

    def test_errors_come_from_the_python_path(self):
        from suchtree_amd import _capi
        for text, exc in (("((A,B);", TreeStructureError), ("(A:x,B);", TreeStructureError), ("", TreeStructureError),
                          ("((A:1,B:1):1);", TypeError), ("(A,B));", TreeStructureError), ("(A,(B,));", TreeStructureError),
                          ("(A:1,B:2]", TreeStructureError), ("(A,'B);", TreeStructureError)):
            assert _capi.newick_native(text) is None, text
            with pytest.raises(exc):
                flat_tree_from_newick(text)

    def test_random_trees_with_random_syntax(self):
        from suchtree_amd.newick import _flat_tree_from_newick_py
        rng = np.random.default_rng(12)

        def gen(depth):
            if depth == 0 or rng.random() < 0.3:
                name = "t%d" % rng.integers(0, 10**6)
                if rng.random() < 0.2:
                    name = "'%s x''y'" % name
                core = name
            else:
                k = int(rng.choice([2, 2, 2, 3, 4, 5]))
                core = "(" + ",".join(gen(depth - 1) for _ in range(k)) + ")"
                r = rng.random()
                if r < 0.3:
                    core += "%.3f" % rng.random()
                elif r < 0.4:
                    core += "lab%d" % rng.integers(0, 99)
            r = rng.random()
            if r < 0.7:
                core += ":%r" % float(np.round(rng.random() * 3, int(rng.integers(1, 9))))
            elif r < 0.8:
                core += ":0"
            elif r < 0.85:
                core += ":%.2e" % (rng.random() * 1e-3)
            if rng.random() < 0.1:
                core += "[c%d]" % rng.integers(0, 9)
            if rng.random() < 0.1:
                core = " " + core + "\n"
            return core

        for _ in range(60):
            text = "(" + gen(6) + "," + gen(6) + ");"
            _same_tree(flat_tree_from_newick(text), _flat_tree_from_newick_py(text))

    def test_big_tree_round_trip(self):
        parent, dist = synth.random_binary_tree(30_000, seed=2, zero_fraction=0.05)
        t = flat_tree_from_newick(synth.to_newick(parent, dist))
        # epsilon edges print as 2.22e-16 and come back as the same float32
        assert np.array_equal(t.parent, parent) and np.array_equal(t.distance, dist)


def test_reference_data_digests():
    """Every tree file under the reference's data/ directory (327 Newick files) through both
    parsers: same arrays from both, and the same arrays as when tests/golden/newick_digests.json
    was written (scripts/newick_digests.py).  Runs where the reference checkout exists."""
    import hashlib
    import json
    ref = "/root/reference/data"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    want = json.load(open(golden_path("newick_digests.json")))
    assert len(want) == 327 and all(v.get("native") == "identical" for v in want.values())
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]   # noqa: E731
    # the big trees, the one with a quote inside an unquoted label, and a spread of the rest
    names = sorted(want)
    sample = set(names[::9]) | {"bigtrees/ml.tree", "bigtrees/nj.tree", "plant-pollinators/rabr/plant.tree"}
    for rel in sorted(sample):
        text = open(os.path.join(ref, rel)).read()
        for native in (False, True):
            t = flat_tree_from_newick(text, native=native)
            w = want[rel]
            assert (t.size, t.num_leaves, t.root, t.depth) == (w["nodes"], w["leaves"], w["root"], w["depth"]), rel
            assert sha(t.parent) == w["parent"] and sha(t.distance) == w["distance"], rel
            assert hashlib.sha256("\n".join(t.leaves.keys()).encode()).hexdigest()[:24] == w["leaf_names"], rel


def test_quote_inside_an_unquoted_label_is_an_ordinary_character():
    """data/plant-pollinators/rabr/plant.tree carries the label Cuphea_o'donellii: a quote opens
    a quoted label only at the start of a token (dendropy's tokenizer)."""
    for native in (False, True):
        t = flat_tree_from_newick("((A_o'b:1,C:2):1,'D e''f':3);", native=native)
        assert list(t.leaves) == ["A_o'b", "C", "D e'f"]
