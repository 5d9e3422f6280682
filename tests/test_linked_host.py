"""SuchLinkedTrees host bookkeeping (link table, link list, subsetting, adjacency /
Laplacian assembly) against the answers printed in the reference's notebooks, with the
oracle standing in for the distance step (no GPU here)."""
import json

import numpy as np
import pandas as pd
import pytest
from scipy.stats import pearsonr

from conftest import golden_path
from oracle.oracle import OracleTree, linked_pairs
from suchtree_amd import SuchTree
from suchtree_amd.linked import SuchLinkedTrees

KNOWN = json.load(open(golden_path("known_answers.json")))


def _gopher_louse():
    d = golden_path("gopher_louse")
    links = pd.read_csv(d + "/links.csv", index_col=0)
    return SuchLinkedTrees(SuchTree(d + "/gopher.tree"), SuchTree(d + "/lice.tree"), links), links


def _fish_worm():
    d = golden_path("fish_worm")
    links = pd.read_csv(d + "/links.csv", index_col=0)
    return SuchLinkedTrees(SuchTree(d + "/host.tree"), SuchTree(d + "/guest.tree"), links), links


def test_gopher_louse_linklist_matches_notebook():
    SLT, links = _gopher_louse()
    assert SLT.n_links == 17 and (SLT.n_rows, SLT.n_cols) == (15, 17)
    got = set(map(tuple, SLT.linklist.tolist()))
    assert got == set(map(tuple, KNOWN["gopher_louse_linklist"]["value"]))
    # column 0 = TreeB (louse) leaf ids in TreeB leaf order, column 1 = TreeA (gopher) leaf ids
    assert SLT.linklist[:, 0].tolist() == sorted(SLT.linklist[:, 0].tolist())
    assert set(SLT.linklist[:, 1]) <= set(SLT.TreeA.leaves.values())
    assert SLT.linkmatrix.sum() == 17 and SLT.linkmatrix.shape == (15, 17)
    lm = links.loc[SLT.row_names, SLT.col_names].to_numpy() > 0
    assert np.array_equal(SLT.linkmatrix, lm)


def test_gopher_louse_linked_distances_statistics_match_notebook():
    """pearsonr over the 136 link pairs printed by the reference: 0.490184989...  The
    notebook predates the float32 left-to-right accumulator (its docstring still describes
    root-distance differences), so the last digits differ; 1e-6 is the stated bar."""
    SLT, _ = _gopher_louse()
    ids_a, ids_b = linked_pairs(SLT.linklist)
    assert len(ids_a) == 136
    OA = OracleTree(SLT.TreeA._flat.parent, SLT.TreeA._flat.distance)
    OB = OracleTree(SLT.TreeB._flat.parent, SLT.TreeB._flat.distance)
    r = pearsonr(OA.distances(ids_a), OB.distances(ids_b))[0]
    assert abs(r - KNOWN["gopher_louse_linked_distances"]["pearson_r"]) < 1e-6


def test_fish_worm_sizes_and_laplacian_shape():
    SLT, _ = _fish_worm()
    k = KNOWN["fish_worm_sizes"]
    assert (SLT.n_links, SLT.TreeA.num_leaves, SLT.TreeB.num_leaves) == (k["links"], k["hosts"], k["guests"])
    aj = SLT.adjacency(on_gpu=False)
    assert aj.shape == (41 + 381, 41 + 381)
    assert np.allclose(aj, aj.T) and aj.max() == pytest.approx(1.0)
    lp = SLT.laplacian(on_gpu=False)
    assert np.allclose(lp.sum(axis=0), 0) and np.allclose(np.diag(lp), aj.sum(axis=0))
    # tree blocks carry 40 + 380 edges, the off-diagonal block the 191 links
    assert (aj[:41, :41] > 0).sum() == 2 * 40 and (aj[41:, 41:] > 0).sum() == 2 * 380
    assert (aj[41:, :41] > 0).sum() == 191
    ev = SLT.spectrum(on_gpu=False)
    assert ev.shape == (422,) and abs(ev[0]) < 1e-9
    # LAPACK's dsyev with the reference's arguments (pyx:3165: 'N', 'U', lwork = 6 N), ascending eigenvalues
    from scipy.linalg.lapack import dsyev
    w, _, info = dsyev(lp, compute_v=0, lower=0, lwork=6 * 422)
    assert info == 0 and np.array_equal(ev, w) and np.all(np.diff(ev) >= 0)
    assert np.allclose(ev, np.linalg.eigvalsh(lp), atol=1e-10)
    # the oracle's dense-block restatement of MuchTree.pyx:1750-1813 + 3081-3145 (no shared code)
    from oracle.oracle import linked_adjacency, linked_laplacian
    fa, fb = SLT.TreeA._flat, SLT.TreeB._flat
    aj_o = linked_adjacency((fa.parent, fa.left, fa.right, fa.distance), (fb.parent, fb.left, fb.right, fb.distance),
                            SLT.linklist, SLT.subset_a_root, SLT.subset_b_root,
                            SLT.TreeA.polytomy_epsilon, SLT.TreeB.polytomy_epsilon)
    assert np.array_equal(aj.view(np.int64), aj_o.view(np.int64))
    assert np.array_equal(lp.view(np.int64), linked_laplacian(aj_o).view(np.int64))


def test_subsetting():
    SLT, links = _gopher_louse()
    A = SLT.TreeA
    node = A.get_parent(A.leaves["Oche"])
    node = A.get_parent(node)                      # a clade with several gophers
    SLT.subset_a(node)
    leaves = set(SLT.subset_a_leafs.tolist())
    assert SLT.subset_a_root == node and SLT.subset_a_size == len(leaves) >= 3
    assert all(int(x) in leaves for x in SLT.linklist[:, 1])
    assert SLT.subset_n_links == sum(1 for b, a in KNOWN["gopher_louse_linklist"]["value"] if a in leaves)
    SLT.subset_a(A.root_node)
    assert SLT.subset_n_links == 17
    B = SLT.TreeB
    SLT.subset_b(B.get_parent(B.leaf_node_ids[0]))
    assert SLT.subset_b_size == 2 and SLT.subset_n_links == 2


def test_constructor_checks():
    SLT, links = _gopher_louse()
    with pytest.raises(Exception, match="link_matrix shape"):
        SuchLinkedTrees(SLT.TreeA, SLT.TreeB, links.iloc[:, :5])
    with pytest.raises(Exception, match="unknown input"):
        SuchLinkedTrees(3, SLT.TreeB, links)
    bad = links.rename(index={links.index[0]: "nobody"})
    with pytest.raises(Exception, match="TreeA leaf names"):
        SuchLinkedTrees(SLT.TreeA, SLT.TreeB, bad)


def test_link_pair_draws_and_bucket_moments_of_the_sampler():
    """The host helpers of sample_linked_distances (st_link_sample_pairs, st_bucket_moments) against the oracle's
    pure-Python restatement of the reference's generator (MuchTree.pyx:2937-2949) and loops (:3025-3048)."""
    import math
    from oracle.oracle import xorshift64star
    from suchtree_amd import _capi
    ll = np.array([[3 * i + 1, 2 * i] for i in range(17)], dtype=np.int64)
    for seed in (1, 12345678901234567, 2 ** 63 - 1, 2 ** 64 - 1, 0):
        qa, qb, state = _capi.link_sample_pairs(seed, ll, 500)
        s = seed
        for k in range(500):
            s, l1 = xorshift64star(s, len(ll))
            s, l2 = xorshift64star(s, len(ll))
            assert (qa[k, 0], qa[k, 1], qb[k, 0], qb[k, 1]) == (ll[l1, 1], ll[l2, 1], ll[l1, 0], ll[l2, 0])
        assert state == s
        qa2, _, state2 = _capi.link_sample_pairs(state, ll, 3)      # continues the sequence
        s2, l1 = xorshift64star(s, len(ll))
        assert qa2[0, 0] == ll[l1, 1] and state2 != state or seed == 0
    d = np.random.default_rng(0).uniform(0, 3, (4, 100)).astype(np.float32).astype(np.float64)
    sums, sumsq = np.zeros(4), np.zeros(4)
    want_s, want_q = [0.0] * 4, [0.0] * 4
    for _ in range(2):      # the running values carry over from cycle to cycle
        _capi.bucket_moments(d, sums, sumsq)
        for i in range(4):
            for j in range(100):
                want_s[i] += d[i, j]
                want_q[i] += math.pow(d[i, j], 2.0)
    assert sums.tolist() == want_s and sumsq.tolist() == want_q
    with pytest.raises(Exception):
        _capi.link_sample_pairs(1, np.zeros((0, 2), dtype=np.int64), 4)


def test_leftover_surface_of_the_reference(capsys):
    """link_leaf / get_links (MuchTree.pyx:1993-2014, used by SuchLinkedTrees.__init__ :2639), dump_table (:3200-3208),
    to_igraph without igraph (:3180-3181), dump_array (:2231-2240), the deprecated quartet wrappers (:2470-2478)."""
    SLT, links = _gopher_louse()
    B = SLT.TreeB
    cols = B.get_links(list(B.leaves.values()))
    assert cols.tolist() == list(range(B.num_leaves)) and cols.dtype == np.dtype(int)
    assert SLT.TreeA.get_links(list(SLT.TreeA.leaves.values())[:3]).tolist() == [-1, -1, -1]      # never linked: the right child
    with pytest.raises(Exception, match="Unknown leaf id"):
        B.get_links([B.root_node])
    with pytest.raises(Exception, match="Cannot link non-leaf node"):
        B.link_leaf(B.root_node, 0)
    SLT.dump_table()
    out = capsys.readouterr().out.splitlines()
    assert len(out) == SLT.n_cols and out[0].startswith("column 0 :")
    assert sum(len(line.split(":")[1].strip().split(",")) for line in out if line.split(":")[1].strip()) == SLT.n_links
    try:
        import igraph      # noqa: F401
    except ImportError:
        with pytest.raises(Exception, match="igraph package not installed."):
            SLT.to_igraph()
    T = SuchTree("(A:1,B:2,(C:3,D:4):5);")
    T.dump_array()
    out = capsys.readouterr().out.splitlines()
    assert len(out) == 5 * T.size and out[0] == "id : 0 ->" and out[1] == "   distance    : 3.000" and out[2] == "   parent      : 1"
