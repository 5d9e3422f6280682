"""Host logic of the SuchTree facade: constructor dispatch, properties,
deprecated aliases, input validation and the error contract -- everything the
reference does in Python before its kernel runs (MuchTree.pyx:872-909, 945-979,
2255-2300, 2374-2459; tests/test_new_api.py:139-158, 810-861).
No distance or MRCA is computed here: without a GPU the product refuses to.
"""
import warnings

import numpy as np
import pytest

from conftest import golden_path
from suchtree_amd import (HipBackendError, InvalidNodeError, NodeNotFoundError, SuchTree,
                          SuchTreeError, _capi)
from suchtree_amd import synth

TEST_TREE = golden_path("test.tree")


@pytest.fixture(scope="module")
def T():
    return SuchTree(TEST_TREE)


def test_constructor_dispatch(T):
    assert T.size == 29 and T.num_leaves == 15 and T.depth == 9 and T.root_node == 25
    S = SuchTree("(A,B,(C,D));")
    assert S.leaves == {"C": 0, "D": 2, "A": 4, "B": 6}
    parent, dist = synth.balanced_tree(4)
    F = SuchTree((parent, dist))
    assert F.size == 31 and F.leaf_names[:2] == ["L0", "L1"] and F.leaves["L1"] == 2
    with pytest.raises(FileNotFoundError):
        SuchTree("no/such/file.tree")
    with pytest.raises(TypeError):
        SuchTree(42)


def test_properties(T):
    assert isinstance(T.size, int) and isinstance(T.depth, int)
    assert all(isinstance(k, str) and isinstance(v, int) for k, v in T.leaves.items())
    assert T.leaf_nodes == {v: k for k, v in T.leaves.items()}
    assert sorted(T.all_nodes.tolist()) == list(range(T.size))
    assert T.leaf_node_ids.tolist() == list(T.leaves.values())
    assert T.leaf_names == list(T.leaves.keys())
    assert len(T.internal_nodes) == T.size - T.num_leaves
    assert T.polytomy_epsilon == np.finfo(np.float64).eps
    assert T.get_parent(T.root_node) == -1 and T.get_children("Ttal") == (-1, -1)
    assert T.get_children(27) == (26, 28) and T.is_leaf(26) and not T.is_leaf(27)
    assert list(T.get_ancestors("Ttal")) == [27, 25]


@pytest.mark.parametrize("old,new", [("length", "size"), ("leafs", "leaves"), ("leafnodes", "leaf_nodes"),
                                     ("n_leafs", "num_leaves"), ("root", "root_node")])
def test_deprecated_properties_warn_and_forward(T, old, new):
    with pytest.warns(DeprecationWarning, match="%s property is deprecated and will be removed in SuchTree 2.0. "
                                                "Use %s instead." % (old, new)):
        assert getattr(T, old) == getattr(T, new)


def test_validate_node(T):
    assert T._validate_node("Ttal") == 26 and T._validate_node(np.int32(3)) == 3
    with pytest.raises(NodeNotFoundError, match="Leaf name not found: nope."):
        T._validate_node("nope")
    for bad in (-1, T.size, T.size + 100):
        with pytest.raises(InvalidNodeError) as e:
            T._validate_node(bad)
        assert str(e.value) == "Node ID %d out of bounds (tree size: 29)" % bad
        assert e.value.node_id == bad and e.value.tree_size == 29
    for bad in (1.5, [1], None):
        with pytest.raises(TypeError, match="Node must be int or str"):
            T._validate_node(bad)
    assert issubclass(InvalidNodeError, SuchTreeError) and issubclass(NodeNotFoundError, SuchTreeError)


def test_pair_array_conventions(T):
    with pytest.raises(ValueError, match=r"Expected \(n, 2\) array, got shape \(1, 3\)"):
        T.distances_bulk(np.array([[0, 1, 2]]))
    with pytest.raises(IndexError):
        T.distances_bulk(np.array([0, 1, 2]))                 # 1-D: the reference's message formatting fails
    with pytest.raises(ValueError):
        T.distances_bulk(np.zeros((0, 2), dtype=np.int64))    # empty: pairs.max() has no identity
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        T.distances_bulk(np.zeros((3, 2), dtype=np.float64))
    c = T._coerce_pairs([(0, 2), (4, 6)])
    assert c.dtype == np.int64 and c.shape == (2, 2)
    assert T._coerce_pairs(np.array([[0, 2]], dtype=np.int32)).dtype == np.int32    # taken as it is
    assert T._coerce_pairs(np.array([[0, 2]], dtype=np.int16)).dtype == np.int64    # widened


def test_nearest_neighbors_of_an_empty_candidate_list_fails_like_the_reference(T):
    """The reference builds np.array([], dtype=int64) from the empty pair list and hands it to
    distances_bulk, whose shape message indexes shape[1] of a 1-D array (MuchTree.pyx:1071-1072, 892-894)."""
    with pytest.raises(IndexError):
        T.nearest_neighbors(0, k=1, from_nodes=[])
    with pytest.raises(ValueError, match="k must be positive"):
        T.nearest_neighbors(0, k=0, from_nodes=[])


def test_by_name_errors_come_before_any_device_work(T):
    with pytest.raises(TypeError, match="pairs must be a list of tuples"):
        T.distances_by_name((("Ttal", "Tbot"),))
    with pytest.raises(TypeError, match="Pair 1: both elements must be strings"):
        T.distances_by_name([("Ttal", "Tbot"), ("Ttal", 3)])
    with pytest.raises(NodeNotFoundError, match="Leaf name not found: Nope."):
        T.distances_by_name([("Ttal", "Nope")])
    with pytest.raises(NodeNotFoundError):
        T.distance("Nope", "Ttal")
    with pytest.raises(InvalidNodeError):
        T.common_ancestor(0, 1000)


def test_no_gpu_means_loud_failure_not_a_cpu_answer(T):
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    for call in (lambda: T.distances_bulk(np.array([[0, 2]])), lambda: T.distance(0, 2),
                 lambda: T.common_ancestor(0, 2), lambda: T.distances_by_name([("Ttal", "Tbot")]),
                 lambda: T.common_ancestors_bulk(np.array([[0, 2]]))):
        with pytest.raises(HipBackendError):
            call()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        with pytest.raises(HipBackendError):
            T.distances(np.array([[0, 2]]))


def test_missing_library_is_reported(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setenv("SUCHTREE_AMD_AUTOBUILD", "0")
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "libsuchtree_hip.so"))
    with pytest.raises(HipBackendError, match="not built"):
        _capi.load()


def test_pickle_round_trip_drops_the_device_handle(T):
    import pickle
    T2 = pickle.loads(pickle.dumps(T))
    assert T2.leaves == T.leaves and T2.size == T.size and T2._dev_tree is None
    assert np.array_equal(T2._flat.parent, T._flat.parent)
    assert "29 nodes" in repr(T2)


def test_names_extension_matches_the_python_loop():
    """csrc/names_ext.c (the C form of the name -> id loop of distances_by_name): same ids as
    the dict lookups; anything unexpected is handed back to the Python loop (-1, no exception)."""
    from suchtree_amd import build as st_build
    st_build.build_names_ext()
    from suchtree_amd import _names
    rng = np.random.default_rng(3)
    leaves = {"leaf_%d" % i: 2 * i for i in range(5000)}
    names = list(leaves)
    pairs = [(names[int(a)], names[int(b)]) for a, b in rng.integers(0, len(names), (20000, 2))]
    out = np.full((len(pairs), 2), -1, dtype=np.int64)
    assert _names.lookup_pairs(pairs, leaves, out) == 0
    assert np.array_equal(out, np.array([(leaves[a], leaves[b]) for a, b in pairs]))
    assert _names.lookup_pairs([], leaves, out) == 0
    for bad in ([("leaf_1", "nope")], [("leaf_1",)], [["leaf_1", "leaf_2"]], [("leaf_1", 3)], [("leaf_1", "leaf_2", "leaf_3")],
                [("leaf_1", "leaf_2"), None]):
        assert _names.lookup_pairs(bad, leaves, out) == -1
    assert _names.lookup_pairs([("a", "b")], {"a": 1, "b": "x"}, out) == -1          # value that is not an int
    with pytest.raises(ValueError):
        _names.lookup_pairs(pairs, leaves, np.empty(10, dtype=np.int64))              # buffer too small
    with pytest.raises(TypeError):
        _names.lookup_pairs(tuple(pairs), leaves, out)                                # not a list


def test_surface_is_complete():
    """Every public method / property name of the reference's SuchTree and SuchLinkedTrees (the names listed here were
    read off SuchTree/MuchTree.pyx; the file itself does not travel) exists on the facade's classes."""
    from suchtree_amd import SuchLinkedTrees
    tree_names = """RED adjacency adjacency_matrix all_nodes bipartition bipartitions common_ancestor degree_sequence depth
        distance distance_matrix distance_to_root distances distances_bulk distances_by_name dump_array edges_data
        get_ancestors get_bipartition get_children get_descendant_nodes get_descendants get_distance_to_root
        get_internal_nodes get_leafs get_leaves get_lineage get_links get_nodes get_parent get_quartet_topology
        get_support has_children has_parent in_order incidence_matrix internal_nodes is_ancestor is_descendant
        is_internal is_internal_node is_leaf is_root is_sibling laplacian laplacian_matrix leaf_names leaf_node_ids
        leaf_nodes leafnodes leafs leaves length link_leaf mrca n_leafs nearest_neighbors nodes_data num_leaves
        pairwise_distances path_between_nodes polytomy_distance polytomy_epsilon pre_order quartet_topologies
        quartet_topologies_bulk quartet_topologies_by_name quartet_topology relationships
        relative_evolutionary_divergence root root_node size to_networkx_edges to_networkx_graph to_networkx_nodes
        to_newick traverse_inorder traverse_internal_only traverse_leaves_only traverse_levelorder traverse_postorder
        traverse_preorder traverse_with_depth traverse_with_distances""".split()
    slt_names = """TreeA TreeB adjacency col_ids col_names dump_table get_column_leafs get_column_links laplacian
        linked_distances linklist linkmatrix n_cols n_links n_rows row_ids row_names sample_linked_distances spectrum
        subset_a subset_a_leafs subset_a_root subset_a_size subset_b subset_b_leafs subset_b_root subset_b_size
        subset_columns subset_n_links to_igraph""".split()
    assert [x for x in tree_names if not hasattr(SuchTree, x)] == []
    assert [x for x in slt_names if not hasattr(SuchLinkedTrees, x)] == []


def test_node_supports_from_the_reference_fixtures():
    """SuchTree/tests/test_SuchTree.py:90-108 and test_new_api.py:229-242 on the reference's own three fixtures: supports
    written as integers and floats are read (float32), supports in comments are dropped with the comments (-1)."""
    for name, want in (("support_int.tree", {1.0, 2.0, 3.0}), ("support_float.tree", {1.342, 2.883, 3.123}), ("support_comment.tree", set())):
        S = SuchTree(golden_path(name))
        assert S.size == 11 and S.num_leaves == 6
        got = {round(S.get_support(int(x)), 3) for x in S.internal_nodes if S.get_support(int(x)) != -1}
        assert got == want, name
        for x in S.get_nodes():
            assert S.get_support(int(x)) != 0 and isinstance(S.get_support(int(x)), float)
        for leaf in S.leaves.values():
            assert S.get_support(leaf) == -1
        if want:      # every internal node on a real (non-epsilon) branch carries its support
            for x in S.internal_nodes:
                x = int(x)
                if x != S.root_node and float(S._flat.distance[x]) != pytest.approx(S.polytomy_epsilon):
                    assert S.get_support(x) != -1.0
    F = SuchTree(golden_path("support_float.tree"))
    assert F.get_support(F.get_parent("C")) == float(np.float32(1.342)) and F.get_parent("C") == F.get_parent("D")
