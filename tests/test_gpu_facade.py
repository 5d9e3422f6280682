"""The reference's own tests for this path, restated against the facade
(SuchTree/tests/test_SuchTree.py:56-88, 163-183; tests/test_new_api.py:362-408,
454-468) -- same fixtures, same assertions, plus exact equality with the oracle."""
import warnings

import numpy as np
import pytest
from pytest import approx

from conftest import golden_path
from oracle.oracle import OracleTree
from suchtree_amd import SuchTree

pytestmark = pytest.mark.gpu

TEST_TREE = golden_path("test.tree")


def _rows():
    for line in open(golden_path("test.matrix")):
        a, b, d = line.split()
        yield a, b, float(d)


def test_distance():
    T = SuchTree(TEST_TREE)
    for a, b, d1 in _rows():
        assert d1 == approx(T.distance(a, b), 0.001)
        assert T.distance(a, b) == T.distance(T.leaves[a], T.leaves[b])


def test_distances_deprecated_alias():
    T = SuchTree(TEST_TREE)
    ids = np.array([(T.leaves[a], T.leaves[b]) for a, b, _ in _rows()], dtype=np.int64)
    with pytest.warns(DeprecationWarning, match=r"distances\(\) is deprecated and will be removed in SuchTree 2.0. "
                                                r"Use distances_bulk\(\) instead."):
        result = T.distances(ids)
    assert isinstance(result, np.ndarray) and result.dtype == np.float64
    for (_, _, d1), d2 in zip(_rows(), result):
        assert d1 == approx(d2, 0.001)


def test_distances_by_name_and_bulk_consistency():
    T = SuchTree(TEST_TREE)
    names = [(a, b) for a, b, _ in _rows()]
    result = T.distances_by_name(names)
    assert isinstance(result, list) and isinstance(result[0], float)
    for (_, _, d1), d2 in zip(_rows(), result):
        assert d1 == approx(d2, 0.001)
    ids = np.array([(T.leaves[a], T.leaves[b]) for a, b in names])
    assert T.distances_bulk(ids).tolist() == result


def test_mrca_by_id_and_name():
    T = SuchTree(TEST_TREE)
    O = OracleTree(T._flat.parent, T._flat.distance)
    leaves = list(T.leaves.items())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        for na, a in leaves:
            for nb, b in leaves:
                m = T.common_ancestor(a, b)
                assert m == T.common_ancestor(b, a) == T.common_ancestor(na, nb) == T.mrca(a, b)
                assert m == O.mrca(a, b)
                anc_a = [a] + list(T.get_ancestors(a))
                anc_b = [b] + list(T.get_ancestors(b))
                assert m in anc_a and m in anc_b
    with pytest.warns(DeprecationWarning, match=r"mrca\(\) is deprecated"):
        T.mrca(0, 2)


def test_docs_known_answers_on_the_gpu():
    T = SuchTree(golden_path("host.tree"))
    assert "%f" % T.distance(12, 26) == "0.388425"
    assert "%f" % T.distance("Reganochromis_calliurus", "Haplotaxodon_microlepis") == "0.270743"
    info = T.device_info()
    assert info["n_nodes"] == 27 and info["n_leaves"] == 14
