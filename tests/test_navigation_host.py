"""Tree navigation of the facade (suchtree_amd/navigate.py), host side: the reference's own assertions for these
methods (SuchTree/tests/test_new_api.py:192-345, 470-494, 562-656, 821-839; tests/test_SuchTree.py:95-160) restated
against the facade on the same fixtures, plus exact orders of enumeration against the oracle's restatements.
Nothing here needs a GPU (no distance, no MRCA)."""
import warnings

import numpy as np
import pytest

from conftest import golden_path
from oracle import oracle
from suchtree_amd import InvalidNodeError, NodeNotFoundError, SuchTree

TEST_TREE = golden_path("test.tree")
SIMPLE = "(A,B,(C,D));"


@pytest.fixture(scope="module")
def T():
    return SuchTree(TEST_TREE)


@pytest.fixture(scope="module")
def S():
    return SuchTree(SIMPLE)


def test_ancestors_and_is_ancestor(T):
    leaf = list(T.leaves.keys())[0]
    ancestors = list(T.get_ancestors(leaf))
    for x in ancestors:
        assert T.is_ancestor(x, leaf) == 1 and T.is_ancestor(leaf, x) == -1
    with pytest.warns(DeprecationWarning, match=r"get_lineage\(\) is deprecated"):
        assert list(T.get_lineage(leaf)) == ancestors
    root = T.root_node
    for x in T.all_nodes:
        if x != root:
            assert T.is_ancestor(root, x) == 1 and T.is_ancestor(x, root) == -1
            assert T.is_descendant(x, root) and not T.is_descendant(root, x)
            assert not T.is_root(x) and T.has_parent(x)
    assert T.is_root(root) and not T.has_parent(root)
    a, b = list(T.leaves.values())[:2]
    assert T.is_ancestor(a, b) == 0 and T.is_ancestor(a, a) == 0
    assert T.is_ancestor("Ttal", 27) == -1 and T.is_ancestor(27, "Ttal") == 1


def test_descendants_leaves_nodes(T):
    root = T.root_node
    assert set(T.get_descendants(root)) == set(T.all_nodes)
    with pytest.warns(DeprecationWarning):
        assert set(T.get_descendant_nodes(root)) == set(T.all_nodes)
    leaves = T.get_leaves(root)
    assert isinstance(leaves, np.ndarray) and set(leaves) == set(T.leaves.values())
    with pytest.warns(DeprecationWarning, match=r"get_leafs\(\) is deprecated"):
        np.testing.assert_array_equal(T.get_leafs(root), leaves)
    assert set(T.get_internal_nodes()) == set(T.internal_nodes) and set(T.get_nodes()) == set(T.all_nodes)
    assert set(T.get_internal_nodes(None)) == set(T.internal_nodes)
    f = T._flat
    for x in range(T.size):      # exact order of enumeration: the oracle's restatement of the to_visit loops
        assert T.get_leaves(x).tolist() == oracle.leaves_below(f.left, f.right, x)
        sub = list(T.get_descendants(x))
        assert T.get_nodes(x).tolist() == sub and sub[0] == x
        assert T.get_internal_nodes(x).tolist() == [y for y in sub if f.left[y] != -1]
    assert T.get_leaves("Ttal").tolist() == [T.leaves["Ttal"]]
    # a 15-leaf tree by hand: node 27 = (Ttal, Tbot)
    assert T.get_leaves(27).tolist() == [26, 28] and list(T.get_descendants(27)) == [27, 26, 28]


def test_node_tests(T, S):
    for x in T.all_nodes:
        assert T.has_children(x) == T.is_internal(x) == (not T.is_leaf(x))
    with pytest.warns(DeprecationWarning, match=r"is_internal_node\(\) is deprecated"):
        assert T.is_internal_node(T.root_node)
    pairs = [(a, b) for a in S.all_nodes for b in S.all_nodes
             if a != b and S.get_parent(a) == S.get_parent(b) and S.get_parent(a) != -1]
    assert pairs and all(S.is_sibling(a, b) for a, b in pairs)
    assert not S.is_sibling(S.root_node, 0) and not S.is_sibling(0, 4) and S.is_sibling("C", "D")
    with pytest.raises(NodeNotFoundError):
        T.is_internal("nope")
    with pytest.raises(InvalidNodeError):
        T.is_root(T.size)


def test_traversals(T, S):
    f = T._flat
    with_d = list(T.traverse_inorder(include_distances=True))
    with pytest.warns(DeprecationWarning, match=r"in_order\(\) is deprecated"):
        assert list(T.in_order(distances=True)) == with_d
    assert len(with_d) == T.size
    assert [x for x, _ in with_d] == list(range(T.size))      # ids ARE in-order positions (MuchTree.pyx:171-216)
    assert all(isinstance(d, float) and d == float(f.distance[x]) for x, d in with_d)
    no_d = list(T.traverse_inorder(include_distances=False))
    assert no_d == list(range(T.size)) and all(isinstance(x, int) for x in no_d)
    pre = list(T.traverse_preorder())
    with pytest.warns(DeprecationWarning, match=r"pre_order\(\) is deprecated"):
        assert list(T.pre_order()) == pre
    assert pre == oracle.preorder(f.left, f.right, T.root_node) and pre[0] == T.root_node and len(pre) == T.size
    post = list(T.traverse_postorder())
    assert len(post) == T.size and post[-1] == T.root_node and all(isinstance(x, int) for x in post)
    seen = set()
    for x in post:      # children before parents, left subtree before right
        l, r = T.get_children(x)
        assert l == -1 or (l in seen and r in seen)
        seen.add(x)
    level = list(T.traverse_levelorder())
    assert level == list(T.get_descendants(T.root_node)) and level[0] == T.root_node
    leaves = list(T.traverse_leaves_only())
    assert leaves == [x for x in pre if T.is_leaf(x)] and set(leaves) == set(T.leaves.values())
    internal = list(T.traverse_internal_only())
    assert internal == [x for x in pre if not T.is_leaf(x)] and set(internal) == set(T.internal_nodes)
    depths = dict(T.traverse_with_depth())
    assert depths[T.root_node] == 0 and len(depths) == T.size
    assert all(isinstance(x, int) and isinstance(d, int) and d == len(list(T.get_ancestors(x))) for x, d in depths.items())
    assert max(depths.values()) + 1 == T.depth
    for x, to_parent, to_root in T.traverse_with_distances():
        assert isinstance(x, int) and isinstance(to_parent, float) and isinstance(to_root, float) and to_root >= 0
        assert to_parent == float(f.distance[x])
        up = [float(f.distance[y]) for y in T.get_ancestors(x)][:-1]      # branches above x, the root's -1 left out
        assert to_root == pytest.approx(sum(up), abs=1e-12)
    # subtrees
    assert list(S.traverse_preorder()) == [3, 1, 0, 2, 5, 4, 6] and list(S.traverse_preorder(5)) == [5, 4, 6]
    assert list(S.traverse_postorder()) == [0, 2, 1, 4, 6, 5, 3] and list(S.traverse_levelorder()) == [3, 1, 5, 0, 2, 4, 6]
    assert list(S.traverse_preorder("C")) == [0] and list(S.traverse_levelorder(1)) == [1, 0, 2]
    assert list(S.traverse_with_depth(1)) == [(1, 0), (0, 1), (2, 1)]


def test_bipartitions(T, S):
    parts = list(T.bipartitions())
    assert len(parts) == len(T.internal_nodes)
    names = set(T.leaves.keys())
    for p in parts:
        assert isinstance(p, frozenset) and len(p) == 2
        a, b = tuple(p)
        assert not (a & b) and (a | b) <= names
    root_part = T.bipartition(T.root_node)
    assert frozenset().union(*root_part) == names
    by_id = T.bipartition(27, by_id=True)
    assert by_id == frozenset((frozenset((26,)), frozenset((28,))))
    assert T.bipartition(27) == frozenset((frozenset(("Ttal",)), frozenset(("Tbot",))))
    with pytest.warns(DeprecationWarning, match=r"get_bipartition\(\) is deprecated"):
        assert T.get_bipartition(27, by_id=True) == by_id
    with pytest.raises(InvalidNodeError, match="Node 26 is not an internal node"):
        T.bipartition(26)
    assert set(S.bipartitions()) == {frozenset((frozenset(("C",)), frozenset(("D",)))),
                                     frozenset((frozenset(("A",)), frozenset(("B",)))),
                                     frozenset((frozenset(("C", "D")), frozenset(("A", "B"))))}


def test_oracle_restatements_on_a_known_tree(S):
    f = S._flat
    # (A,B,(C,D)); -> C 0, D 2, A 4, B 6 (docs: test_new_api.py), root 3
    assert S.root_node == 3
    assert oracle.preorder(f.left, f.right, 3) == list(S.traverse_preorder())
    assert oracle.leaves_below(f.left, f.right, 3) == S.get_leaves(3).tolist()
    assert sorted(S.get_leaves(3).tolist()) == [0, 2, 4, 6]
    with warnings.catch_warnings():
        warnings.simplefilter("error")      # the new names do not warn
        list(S.get_ancestors(0)); S.get_leaves(3); S.is_internal(3); list(S.traverse_preorder())


def test_graph_matrices(T, S):
    f = T._flat
    for start in (None, 27, 19, "Ttal"):      # SuchTree/tests/test_new_api.py:657-706, 729-748 + the oracle's restatement
        r = T.adjacency_matrix(start)
        adj, ids = r["adjacency_matrix"], r["node_ids"]
        want, want_ids = oracle.tree_adjacency(f.parent, f.left, f.right, f.distance,
                                               T.root_node if start is None else T._validate_node(start), T.polytomy_epsilon)
        assert ids.tolist() == want_ids.tolist() and np.array_equal(adj, want)
        assert adj.shape == (len(ids), len(ids)) and np.array_equal(adj, adj.T) and adj.dtype == np.float64
        lap = T.laplacian_matrix(start)
        assert lap["node_ids"].tolist() == ids.tolist()
        assert np.array_equal(lap["laplacian"], np.diag(adj.sum(axis=0)) - adj)
        np.testing.assert_allclose(lap["laplacian"].sum(axis=1), 0, atol=1e-12)
        deg = T.degree_sequence(start)
        assert deg["degrees"].tolist() == (adj > 0).sum(axis=1).tolist()
        assert deg["max_degree"] == deg["degrees"].max() and deg["min_degree"] == deg["degrees"].min()
    assert T.adjacency_matrix()["adjacency_matrix"].shape == (T.size, T.size)
    assert T.degree_sequence()["max_degree"] == 3 and T.degree_sequence()["min_degree"] == 1
    inc = T.incidence_matrix()
    m, ids, edges = inc["incidence_matrix"], inc["node_ids"], inc["edge_list"]
    assert m.shape == (T.size, T.size - 1) and len(edges) == T.size - 1
    assert (m.sum(axis=0) == 0).all() and ((m == 1).sum(axis=0) == 1).all() and ((m == -1).sum(axis=0) == 1).all()
    for k, (p, c) in enumerate(edges):
        assert T.get_parent(c) == p and m[ids.tolist().index(p), k] == 1 and m[ids.tolist().index(c), k] == -1
    with pytest.raises(IndexError):
        T.incidence_matrix(27)
    # zero-length branches count as epsilon (MuchTree.pyx:1801-1802)
    Z = SuchTree("((A:0,B:1):1,C:2);")
    assert Z.adjacency_matrix()["adjacency_matrix"].min() == 0 and \
        np.sort(Z.adjacency_matrix()["adjacency_matrix"][np.triu_indices(5, 1)])[-4] == Z.polytomy_epsilon
    with pytest.warns(DeprecationWarning, match=r"adjacency\(\) is deprecated"):
        assert np.array_equal(T.adjacency()["adjacency_matrix"], T.adjacency_matrix()["adjacency_matrix"])
    with pytest.warns(DeprecationWarning, match=r"laplacian\(\) is deprecated"):
        assert np.array_equal(T.laplacian()["laplacian"], T.laplacian_matrix()["laplacian"])


def test_edges_and_newick_export(T, S):
    # SuchTree/tests/test_new_api.py:767-806
    edges = list(T.to_networkx_edges())
    with pytest.warns(DeprecationWarning, match=r"edges_data\(\) is deprecated"):
        assert list(T.edges_data()) == edges
    assert len(edges) == T.size - 1
    for child, parent, attrs in edges:
        assert isinstance(child, int) and isinstance(parent, int) and isinstance(attrs, dict)
        assert T.get_parent(child) == parent and attrs["weight"] == attrs["length"] == float(T._flat.distance[child])
    text = T.to_newick()
    assert isinstance(text, str) and text.endswith(";") and "(" in text and ")" in text
    R = SuchTree(text)
    assert R.num_leaves == T.num_leaves and R.leaves == T.leaves and R.size == T.size
    assert np.array_equal(R._flat.parent, T._flat.parent) and np.array_equal(R._flat.distance, T._flat.distance)
    assert S.to_newick(include_distances=False) == "((C,D),(A,B));"
    assert S.to_newick() == "((C:2.220446049250313e-16,D:2.220446049250313e-16):2.220446049250313e-16," \
                            "(A:2.220446049250313e-16,B:2.220446049250313e-16):2.220446049250313e-16);"
    assert S.to_newick(1, include_distances=False) == "(C,D);" and S.to_newick("A") == "A;"
    U = SuchTree("((A:1,B:2)0.9:0.5,(C:3,D:4)75:0.25);")
    assert U.to_newick() == "((A:1.0,B:2.0)" + str(float(np.float32(0.9))) + ":0.5,(C:3.0,D:4.0)75.0:0.25);"
    assert U.to_newick(include_support=False) == "((A:1.0,B:2.0):0.5,(C:3.0,D:4.0):0.25);"
    # a caterpillar deeper than Python's recursion limit
    n = 3000
    deep = "(" * (n - 1) + "L0:1" + "".join(",L%d:1):1" % i for i in range(1, n))
    D = SuchTree(deep[:deep.rindex(":")] + ";")
    assert SuchTree(D.to_newick()).num_leaves == n


def test_traversals_and_newick_round_trip_on_a_deep_real_tree(ml_arrays):
    """ml.tree (108,653 nodes, 376 levels): every traversal visits every node once, in the order its name says; the
    Newick export parses back to the same node table (ids are in-order positions, lengths print as float32 values)."""
    parent, dist, leaf_ids = ml_arrays
    names = ["t%d" % i for i in range(len(leaf_ids))]
    T = SuchTree((parent, dist, names))
    f = T._flat
    n = T.size
    pre = np.fromiter(T.traverse_preorder(), dtype=np.int64)
    post = np.fromiter(T.traverse_postorder(), dtype=np.int64)
    level = np.fromiter(T.traverse_levelorder(), dtype=np.int64)
    ino = np.fromiter(T.traverse_inorder(include_distances=False), dtype=np.int64)
    for order in (pre, post, level, ino):
        assert len(order) == n and np.array_equal(np.sort(order), np.arange(n))
    where_pre, where_post = np.empty(n, np.int64), np.empty(n, np.int64)
    where_pre[pre] = np.arange(n)
    where_post[post] = np.arange(n)
    kids = np.flatnonzero(f.parent >= 0)
    assert (where_pre[f.parent[kids]] < where_pre[kids]).all() and (where_post[f.parent[kids]] > where_post[kids]).all()
    depth = dict(T.traverse_with_depth())
    assert max(depth.values()) + 1 == T.depth == 376
    assert [depth[int(x)] for x in level] == sorted(depth.values())
    assert np.array_equal(np.sort(T.get_leaves(T.root_node)), np.sort(leaf_ids))
    assert len(T.get_internal_nodes()) == n - len(leaf_ids)
    text = T.to_newick()
    R = SuchTree(text)
    assert R.size == n and R.num_leaves == T.num_leaves and R.depth == T.depth
    # the parser numbers nodes in order; relabel T's nodes by in-order position and the tables must coincide
    pos = np.empty(n, np.int64)
    pos[ino] = np.arange(n)
    want_parent = np.full(n, -1, np.int64)
    want_parent[pos[kids]] = pos[f.parent[kids]]
    want_dist = np.empty(n, np.float32)
    want_dist[pos] = f.distance
    assert np.array_equal(R._flat.parent, want_parent) and np.array_equal(R._flat.distance.view(np.uint32), want_dist.view(np.uint32))
    assert {R.leaves[name] for name in names[:50]} == {int(pos[T.leaves[name]]) for name in names[:50]}
