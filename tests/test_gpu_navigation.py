"""Callers of the path in suchtree_amd/navigate.py on the GPU: distance_to_root, path_between_nodes and the relative
evolutionary divergence, against the oracle's statement-by-statement restatements (oracle/oracle.py) -- exact
equality -- and the reference's own assertions (SuchTree/tests/test_new_api.py:354-360, 539-556;
tests/test_SuchTree.py:46-49: distance to root of every leaf)."""
import numpy as np
import pytest

from conftest import golden_path
from oracle import oracle
from oracle.oracle import OracleTree
from suchtree_amd import SuchTree

pytestmark = pytest.mark.gpu


def _oracle_of(T):
    f = T._flat
    return OracleTree(f.parent, f.distance, f.left, f.right)


@pytest.mark.parametrize("tree", ["test.tree", "host.tree", "fish_worm/guest.tree"])
def test_distance_to_root_and_paths(tree):
    T = SuchTree(golden_path(tree))
    f = T._flat
    for x in range(T.size):
        d = T.distance_to_root(x)
        assert isinstance(d, float) and d == oracle.distance_to_root(f.parent, f.distance, x)
    for name in T.leaves:
        with pytest.warns(DeprecationWarning, match=r"get_distance_to_root\(\) is deprecated"):
            assert T.get_distance_to_root(name) == T.distance_to_root(name)
    assert T.distance_to_root(T.root_node) == 0.0
    O = _oracle_of(T)
    rng = np.random.default_rng(7)
    for a, b in rng.integers(0, T.size, (60, 2)):
        a, b = int(a), int(b)
        path = T.path_between_nodes(a, b)
        assert isinstance(path, list) and all(isinstance(x, int) for x in path)
        assert path[0] == a and path[-1] == b and len(set(path)) == len(path)
        m = O.mrca(a, b)
        assert m in path
        k = path.index(m)
        assert all(f.parent[path[i]] == path[i + 1] for i in range(k))                    # up to the MRCA
        assert all(f.parent[path[i + 1]] == path[i] for i in range(k, len(path) - 1))     # and down again
    leaves = list(T.leaves.keys())[:2]
    path = T.path_between_nodes(leaves[0], leaves[1])
    assert len(path) >= 2 and path[0] == T.leaves[leaves[0]] and path[-1] == T.leaves[leaves[1]]
    assert T.path_between_nodes(3, 3) == [3]


def test_distance_to_root_stops_at_a_length_of_minus_one():
    # MuchTree.pyx:840-845 ends its loop at the first length equal to -1, not at the root: a branch of length -1
    # (neighbor-joining trees have negative lengths) cuts the sum short there, in the reference and here alike
    T = SuchTree("((A:1,B:2):-1,(C:3,D:4):0.5);")
    f = T._flat
    for x in range(T.size):
        assert T.distance_to_root(x) == oracle.distance_to_root(f.parent, f.distance, x)
    assert T.distance_to_root("A") == 1.0 and T.distance_to_root("C") == 3.5


@pytest.mark.parametrize("tree", ["test.tree", "host.tree", "fish_worm/guest.tree", "gopher_louse/lice.tree"])
def test_relative_evolutionary_divergence(tree):
    T = SuchTree(golden_path(tree))
    f = T._flat
    red = T.relative_evolutionary_divergence
    want = oracle.relative_evolutionary_divergence(_oracle_of(T), f.left, f.right, T.root_node)
    assert list(red.keys()) == list(want.keys())      # filled in pre-order, like the reference's dict
    for x in want:
        assert red[x] == want[x], (x, red[x], want[x])
    # what the reference's docstring states (MuchTree.pyx:306-312): 0 at the root, 1 at every leaf, between
    assert red[T.root_node] == 0 and all(abs(red[x] - 1) < 1e-12 for x in T.leaves.values())
    assert all(-1e-12 <= v <= 1 + 1e-12 for v in red.values())
    assert T.RED is red and T.relative_evolutionary_divergence is red      # cached (MuchTree.pyx:316-318)


def test_relative_evolutionary_divergence_on_a_deep_tree(ml_arrays):
    # ml.tree: 108,653 nodes, 376 levels; the pairs of all nodes with all leaves below them are one 6e6-pair batch
    parent, dist = ml_arrays[0], ml_arrays[1]
    T = SuchTree((parent, dist))
    f = T._flat
    red = T.relative_evolutionary_divergence
    assert len(red) == T.size and red[T.root_node] == 0
    rng = np.random.default_rng(3)
    O = _oracle_of(T)
    # the oracle's restatement on a sample of lineages: RED of a node needs the REDs of its ancestors only
    pre = oracle.preorder(f.left, f.right, T.root_node)
    sample = set()
    for x in rng.choice(T.size, 40, replace=False):
        sample.add(int(x)); sample.update(int(y) for y in T.get_ancestors(int(x)))
    want = {T.root_node: 0}
    for x in pre[1:]:
        if x not in sample:
            continue
        P = want[int(f.parent[x])]
        a = O.distance(x, int(f.parent[x]))
        leaves = oracle.leaves_below(f.left, f.right, x)
        b = np.mean([float(v) for v in O.distances(np.stack((np.full(len(leaves), x), np.array(leaves)), axis=1))])
        want[x] = P + (a / (a + b)) * (1 - P)
    for x, v in want.items():
        assert red[x] == v, (x, red[x], v)


def test_networkx_export_and_bulk_distances_to_root():
    # SuchTree/tests/test_new_api.py:752-790
    T = SuchTree(golden_path("test.tree"))
    f = T._flat
    nodes = list(T.to_networkx_nodes())
    with pytest.warns(DeprecationWarning, match=r"nodes_data\(\) is deprecated"):
        assert list(T.nodes_data()) == nodes
    assert len(nodes) == T.size and [x for x, _ in nodes] == list(T.get_descendants(T.root_node))
    for node_id, attrs in nodes:
        assert isinstance(node_id, int) and isinstance(attrs, dict) and attrs["type"] in ("leaf", "internal")
        assert attrs["type"] == ("leaf" if T.is_leaf(node_id) else "internal")
        assert attrs["label"] == (T.leaf_nodes[node_id] if T.is_leaf(node_id) else "node_%d" % node_id)
        assert attrs["distance_to_root"] == oracle.distance_to_root(f.parent, f.distance, node_id)
        assert attrs["depth"] == len(list(T.get_ancestors(node_id))) and isinstance(attrs["depth"], int)
        assert ("distance_to_parent" in attrs) == (node_id != T.root_node)
        assert list(attrs)[:2] == ["type", "label"] and list(attrs)[-2:] == ["distance_to_root", "depth"]
    all_d = T.distances_to_root_bulk()
    assert all_d.dtype == np.float64 and all_d.tolist() == [oracle.distance_to_root(f.parent, f.distance, x) for x in range(T.size)]
    assert T.distances_to_root_bulk(["Ttal", 27]).tolist() == [T.distance_to_root("Ttal"), T.distance_to_root(27)]
    sub = list(T.to_networkx_nodes(27))
    assert [x for x, _ in sub] == [27, 26, 28] and sub[1][1]["depth"] == 2
    nx = pytest.importorskip("networkx")
    G = T.to_networkx_graph()
    assert isinstance(G, nx.Graph) and len(G.nodes) == T.size and len(G.edges) == T.size - 1
    assert G.nodes[T.root_node]["type"] == "internal" and G.edges[26, 27]["weight"] == float(f.distance[26])
    Z = SuchTree("((A:1,B:2):-1,(C:3,D:4):0.5);")      # a branch of length -1 ends the reference's root-ward loop
    assert Z.distances_to_root_bulk().tolist() == [oracle.distance_to_root(Z._flat.parent, Z._flat.distance, x) for x in range(Z.size)]


def test_relationships_and_deprecated_quartet_names():
    """relationships (MuchTree.pyx:2158-2178): every column against the facade's own single calls and the oracle."""
    T = SuchTree(golden_path("test.tree"))
    f = T._flat
    R = T.relationships()
    n = T.num_leaves
    assert list(R.columns) == ["a", "b", "distance", "a_to_root", "b_to_root", "mrca", "mrca_to_root", "a_to_mrca", "b_to_mrca"]
    assert len(R) == n * (n - 1) // 2 and {frozenset((a, b)) for a, b in zip(R["a"], R["b"])} == \
        {frozenset(p) for p in __import__("itertools").combinations(T.leaves.keys(), 2)}
    O = _oracle_of(T)
    for row in R.itertuples():
        a, b = T.leaves[row.a], T.leaves[row.b]
        assert row.distance == O.distance(a, b) and row.mrca == O.mrca(a, b)
        assert row.a_to_root == oracle.distance_to_root(f.parent, f.distance, a)
        assert row.b_to_root == oracle.distance_to_root(f.parent, f.distance, b)
        assert row.mrca_to_root == oracle.distance_to_root(f.parent, f.distance, row.mrca)
        assert row.a_to_mrca == row.a_to_root - row.mrca_to_root and row.b_to_mrca == row.b_to_root - row.mrca_to_root
    leaves = list(T.leaves.keys())[:4]
    with pytest.warns(DeprecationWarning, match=r"get_quartet_topology\(\) is deprecated"):
        assert T.get_quartet_topology(*leaves) == T.quartet_topology(*leaves)
    q = np.array([[T.leaves[x] for x in leaves]], dtype=np.int64)
    with pytest.warns(DeprecationWarning, match=r"quartet_topologies\(\) is deprecated"):
        assert np.array_equal(T.quartet_topologies(q), T.quartet_topologies_bulk(q))
