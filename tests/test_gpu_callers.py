"""Callers either side of the hot path, on the GPU: the all-pairs generator
(pairwise_distances / distance_matrix / nearest_neighbors; BASELINE config 4 at reduced
size) and SuchLinkedTrees.linked_distances (config 5)."""
import json

import numpy as np
import pandas as pd
import pytest
from scipy.stats import pearsonr

from conftest import assert_bits_equal, golden_path
from oracle.oracle import OracleTree, linked_adjacency, linked_laplacian, linked_pairs
from suchtree_amd import InvalidNodeError, SuchTree, _capi, sharding, synth
from suchtree_amd.linked import SuchLinkedTrees

pytestmark = pytest.mark.gpu
KNOWN = json.load(open(golden_path("known_answers.json")))


def _tri_pairs(ids):
    rows, cols = np.tril_indices(len(ids), -1)
    return np.stack([ids[cols], ids[rows]], axis=1).astype(np.int64)


@pytest.mark.parametrize("strategy", ["walk", "canopy"])
def test_triangle_generator_equals_explicit_pairs(strategy, ml_arrays):
    parent, dist, leaf_ids = ml_arrays
    dev = _capi.DeviceTree(parent, dist)
    dev.set_strategy(strategy)
    O = OracleTree(parent, dist)
    ids = np.random.default_rng(4).choice(leaf_ids, size=700, replace=False)
    pairs = _tri_pairs(ids)
    want_d, want_m = O.distances(pairs), O.mrca_bulk(pairs)
    d, m = dev.triangle_host(ids, want_dist=True, want_mrca=True)
    assert_bits_equal(d, want_d)
    assert np.array_equal(m, want_m)
    # any k-range gives the matching slice (what multi-GPU sharding and tiling rely on)
    total = len(pairs)
    for g in range(3):
        lo, hi = sharding.triangle_shard_bounds(len(ids), 3, g)
        d, m = dev.triangle_host(ids, k_begin=lo, k_count=hi - lo, want_dist=True, want_mrca=True)
        assert_bits_equal(d, want_d[lo:hi])
        assert np.array_equal(m, want_m[lo:hi])
    assert sharding.triangle_shard_bounds(len(ids), 3, 2)[1] == total
    # strided id list (a link-list column), empty and single-id lists
    wide = np.zeros((len(ids), 2), dtype=np.int64)
    wide[:, 1] = ids
    assert_bits_equal(dev.triangle_host(wide[:, 1])[0], want_d)
    assert dev.triangle_host(ids[:1])[0].shape == (0,)
    assert dev.triangle_host(ids[:0])[0].shape == (0,)
    with pytest.raises(InvalidNodeError):
        dev.triangle_host(np.array([0, 2, len(parent) + 1]))
    dev.close()


def test_config4_all_pairs_reduced_size():
    """Config 4 is the full lower triangle of a 100k-leaf tree (5e9 pairs); here a
    6000-leaf slice of that workload (1.8e7 pairs) against the oracle on a sample, plus
    size-independent properties on the whole triangle."""
    parent, dist = synth.complete_tree(100_000, seed=44)
    dev = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    ids = np.arange(0, 12_000, 2, dtype=np.int64) + 40_000          # 6000 consecutive leaves
    d, m = dev.triangle_host(ids, want_dist=True, want_mrca=True)
    assert len(d) == 6000 * 5999 // 2
    pairs = _tri_pairs(ids)
    pick = np.random.default_rng(0).integers(0, len(pairs), 300_000)
    assert_bits_equal(d[pick], O.distances(pairs[pick]))
    assert np.array_equal(m[pick], O.mrca_bulk(pairs[pick]))
    # in-order ids: the MRCA of leaves a < b lies strictly between them
    assert np.all((m > pairs[:, 0]) & (m < pairs[:, 1]))
    dev.set_strategy("walk")
    d2, m2 = dev.triangle_host(ids, want_dist=True, want_mrca=True)
    assert_bits_equal(d2, d)
    assert np.array_equal(m2, m)
    dev.close()


def test_pairwise_distance_matrix_and_neighbors():
    T = SuchTree(golden_path("test.tree"))
    O = OracleTree(T._flat.parent, T._flat.distance)
    D = T.pairwise_distances()
    n = T.num_leaves
    assert D.shape == (n, n) and np.array_equal(D, D.T) and np.all(np.diag(D) == 0)
    ids = T.leaf_node_ids
    for i in range(n):
        for j in range(i + 1, n):
            assert D[i, j] == O.distance(int(ids[i]), int(ids[j]))      # d(ids[i], ids[j]), i < j
    names = {r[0] + r[1]: float(r[2]) for r in (line.split() for line in open(golden_path("test.matrix")))}
    leaf_names = T.leaf_names
    for i in range(n):
        for j in range(n):
            assert D[i, j] == pytest.approx(names[leaf_names[i] + leaf_names[j]], rel=1e-3, abs=1e-9)
    sub = T.pairwise_distances(["Ttal", "Tbot", 27, 3])
    assert sub.shape == (4, 4) and sub[0, 1] == O.distance(26, 28) and sub[2, 3] == O.distance(27, 3)
    assert T.pairwise_distances(["Ttal"]).shape == (1, 1)
    dm = T.distance_matrix(["Ttal", 27])
    assert dm["node_names"] == ["Ttal", "node_27"] and dm["node_ids"].tolist() == [26, 27]
    nn = T.nearest_neighbors("Ttal", k=3)
    want = sorted(((O.distance(26, int(i)), T.leaf_nodes[int(i)]) for i in ids if i != 26))[:3]
    assert [x[0] for x in nn] == [w[1] for w in want] and [x[1] for x in nn] == [w[0] for w in want]
    with pytest.raises(ValueError, match="k must be positive"):
        T.nearest_neighbors("Ttal", k=0)


def test_grid_generator_pairwise_matrix_and_knn_on_a_real_tree(ml_arrays):
    """f2 on the device: the symmetric matrix written by the grid generator (no pair list, no
    host scatter) and the per-row top-k, against the oracle on data/bigtrees/ml.tree."""
    parent, dist, leaf_ids = ml_arrays
    T = SuchTree((parent, dist))
    O = OracleTree(parent, dist)
    rng = np.random.default_rng(31)
    ids = rng.choice(leaf_ids, size=1500, replace=False)
    ids[7] = int(parent[ids[8]])                      # an internal node among them
    D = T.pairwise_distances([int(x) for x in ids])
    n = len(ids)
    assert D.shape == (n, n) and np.array_equal(D, D.T) and np.all(np.diag(D) == 0)
    iu, ju = np.triu_indices(n, 1)                    # the reference's pairs: (ids[i], ids[j]), i < j
    assert_bits_equal(D[iu, ju], O.distances(np.stack([ids[iu], ids[ju]], 1)))
    # rectangular grid, an element sub-range, MRCA ids
    dev = T._device_tree()
    rows, cols = ids[:37], ids[100:1100]
    full_d, full_m = dev.grid_host(rows, cols, want_dist=True, want_mrca=True)
    rr, cc = np.divmod(np.arange(len(rows) * len(cols)), len(cols))
    rect_pairs = np.stack([rows[rr], cols[cc]], 1)
    assert_bits_equal(full_d, O.distances(rect_pairs))
    assert np.array_equal(full_m, O.mrca_bulk(rect_pairs))
    part_d, _ = dev.grid_host(rows, cols, e_begin=12_345, e_count=5000)
    assert_bits_equal(part_d, full_d[12_345:17_345])
    for strategy in ("walk", "canopy"):
        dev.set_strategy(strategy)
        d2, _ = dev.grid_host(ids[:300], ids[:300], symmetric=True)
        assert_bits_equal(d2, D[:300, :300].reshape(-1), strategy)
    dev.set_strategy("auto")
    with pytest.raises(ValueError):
        dev.grid_host(rows, cols, symmetric=True)
    with pytest.raises(InvalidNodeError):
        dev.grid_host(np.array([0, len(parent)]), ids[:5000])
    # k nearest leaves of many queries; selection on the GPU
    queries = ids[:64]
    k = 9
    nb_ids, nb_d = T.nearest_neighbors_bulk([int(q) for q in queries], k=k)
    leaves = np.asarray(T.leaf_node_ids, dtype=np.int64)
    for qi, q in enumerate(queries[:16]):
        cand = leaves[leaves != q]
        row = O.distances(np.stack([np.full(len(cand), q), cand], 1))
        assert_bits_equal(nb_d[qi], np.sort(row)[:k])                           # the k smallest, ascending
        assert len(set(nb_ids[qi].tolist())) == k and q not in nb_ids[qi]
        assert_bits_equal(O.distances(np.stack([np.full(k, q), nb_ids[qi]], 1)), nb_d[qi])   # ids match their distances
        order = np.lexsort((np.arange(len(cand)), row))[:k]                     # ties: first listed first
        assert np.array_equal(cand[order], nb_ids[qi])
    one = T.nearest_neighbors(int(queries[3]), k=k)
    assert [x[0] for x in one] == [T.leaf_nodes[int(i)] for i in nb_ids[3]] and [x[1] for x in one] == nb_d[3].tolist()
    few_ids, few_d = T.nearest_neighbors_bulk([int(queries[0])], k=5, from_nodes=[int(x) for x in ids[:3]])
    assert few_ids[0, 3:].tolist() == [-1, -1] and np.isnan(few_d[0, 3:]).all() and np.isfinite(few_d[0, :3]).all()
    big = T.nearest_neighbors(int(queries[0]), k=300, from_nodes=[int(x) for x in ids])    # beyond the device's k: host sort
    row = O.distances(np.stack([np.full(n, queries[0]), ids], 1))
    assert [x[1] for x in big] == np.sort(row)[:300].tolist()


def test_nearest_neighbors_special_values_order_like_argsort():
    """NaN distances (either sign bit) come last and -0.0 ties with +0.0, as in the reference's np.argsort
    (MuchTree.pyx:1075); the device selection (k <= 256) and the host sort (k > 256) must agree."""
    n_leaves = 400
    parent, dist = synth.caterpillar_tree(n_leaves)
    dist = dist.copy()
    leaves = np.arange(0, 2 * n_leaves, 2)
    dist[leaves[10:20]] = np.float32(np.nan)
    dist[leaves[20:30]] = np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0]      # NaN with the sign bit set
    T = SuchTree((parent, dist))
    cands = [int(x) for x in leaves[5:60]]
    small = T.nearest_neighbors(int(leaves[0]), k=len(cands), from_nodes=cands)
    ds = np.array([d for _, d in small])
    n_nan = int(np.isnan(ds).sum())
    assert n_nan == 20 and not np.isnan(ds[:-n_nan]).any() and np.isnan(ds[-n_nan:]).all()
    assert np.all(np.diff(ds[:-n_nan]) >= 0)
    big = T.nearest_neighbors(int(leaves[0]), k=300, from_nodes=cands)       # host argsort path
    assert [x for x, _ in big[: len(cands) - n_nan]] == [x for x, _ in small[: len(cands) - n_nan]]
    # zero-length and negative-zero-length branches give +0.0 and -0.0 ... as equal keys: index order decides
    parent2, dist2 = synth.balanced_tree(4)
    dist2 = dist2.copy()
    dist2[:] = np.float32(0.0)
    dist2[::4] = np.float32(-0.0)
    Z = SuchTree((parent2, dist2))
    zc = [int(x) for x in range(2, 32, 2)]
    got = Z.nearest_neighbors(0, k=len(zc), from_nodes=zc)
    assert [x for x, _ in got] == zc and all(d == 0.0 for _, d in got)


@pytest.mark.parametrize("which", ["gopher_louse", "fish_worm"])
def test_config5_linked_distances(which):
    d = golden_path(which)
    names = ("gopher.tree", "lice.tree") if which == "gopher_louse" else ("host.tree", "guest.tree")
    links = pd.read_csv(d + "/links.csv", index_col=0)
    SLT = SuchLinkedTrees(SuchTree(d + "/" + names[0]), SuchTree(d + "/" + names[1]), links)
    res = SLT.linked_distances()
    L = SLT.n_links
    assert res["n_pairs"] == res["n_samples"] == L * (L - 1) // 2 == len(res["TreeA"]) == len(res["TreeB"])
    ids_a, ids_b = linked_pairs(SLT.linklist)                       # oracle: MuchTree.pyx:2918-2925
    assert np.array_equal(res["ids_A"], ids_a) and np.array_equal(res["ids_B"], ids_b)
    OA = OracleTree(SLT.TreeA._flat.parent, SLT.TreeA._flat.distance)
    OB = OracleTree(SLT.TreeB._flat.parent, SLT.TreeB._flat.distance)
    assert_bits_equal(res["TreeA"], OA.distances(ids_a))
    assert_bits_equal(res["TreeB"], OB.distances(ids_b))
    # adjacency / Laplacian: dense assembly on the GPU against the oracle's own dense-block
    # restatement of MuchTree.pyx:1750-1813 + 3081-3145 (oracle/oracle.py, no shared code)
    aj_gpu, lp_gpu = SLT.adjacency(), SLT.laplacian()
    fa, fb = SLT.TreeA._flat, SLT.TreeB._flat
    aj_o = linked_adjacency((fa.parent, fa.left, fa.right, fa.distance), (fb.parent, fb.left, fb.right, fb.distance),
                            SLT.linklist, SLT.subset_a_root, SLT.subset_b_root,
                            SLT.TreeA.polytomy_epsilon, SLT.TreeB.polytomy_epsilon)
    lp_o = linked_laplacian(aj_o)
    n_graph = SLT.TreeA.size + SLT.TreeB.size
    assert aj_gpu.shape == lp_gpu.shape == (n_graph, n_graph)
    assert np.array_equal(aj_gpu.view(np.int64), aj_o.view(np.int64))
    assert np.array_equal(lp_gpu.view(np.int64), lp_o.view(np.int64))
    assert np.array_equal(SLT.adjacency(on_gpu=False).view(np.int64), aj_o.view(np.int64))
    assert np.allclose(lp_gpu.sum(axis=0), 0)
    # a subsetted graph (a clade of TreeB and the links into it)
    B = SLT.TreeB          # a clade two levels above some linked leaf that is not the whole tree
    sub_root = next(g for g in (int(B.get_parent(B.get_parent(int(x)))) for x in SLT.linklist[:, 0]
                                if B.get_parent(int(x)) != B.root_node)
                    if g not in (-1, B.root_node))
    SLT.subset_b(sub_root)
    aj_s = linked_adjacency((fa.parent, fa.left, fa.right, fa.distance), (fb.parent, fb.left, fb.right, fb.distance),
                            SLT.linklist, SLT.subset_a_root, SLT.subset_b_root,
                            SLT.TreeA.polytomy_epsilon, SLT.TreeB.polytomy_epsilon)
    assert aj_s.shape[0] < n_graph
    assert np.array_equal(SLT.adjacency().view(np.int64), aj_s.view(np.int64))
    assert np.array_equal(SLT.laplacian().view(np.int64), linked_laplacian(aj_s).view(np.int64))
    SLT.subset_b(SLT.TreeB.root_node)
    if which == "gopher_louse":
        r = pearsonr(res["TreeA"], res["TreeB"])[0]
        assert abs(r - KNOWN["gopher_louse_linked_distances"]["pearson_r"]) < 1e-6
    else:
        assert L == 191 and res["n_pairs"] == 18145


@pytest.mark.parametrize("which", ["gopher_louse", "fish_worm"])
def test_sample_linked_distances_against_the_restatement(which):
    """SuchLinkedTrees.sample_linked_distances (MuchTree.pyx:2951-3079, a caller of distances_bulk at :3039-3040): the
    reference's generator, its draws, its bucket moments and its stop rule -- against oracle.sample_linked_distances,
    the same loops in pure Python over the CPU oracle's distances, from the same generator state.  Everything the
    reference returns must be identical, and the generator must be left in the same state."""
    from oracle.oracle import sample_linked_distances
    d = golden_path(which)
    names = ("gopher.tree", "lice.tree") if which == "gopher_louse" else ("host.tree", "guest.tree")
    links = pd.read_csv(d + "/links.csv", index_col=0)
    SLT = SuchLinkedTrees(SuchTree(d + "/" + names[0]), SuchTree(d + "/" + names[1]), links)
    fa, fb = SLT.TreeA._flat, SLT.TreeB._flat
    OA, OB = OracleTree(fa.parent, fa.distance), OracleTree(fb.parent, fb.distance)
    ll = [tuple(int(x) for x in row) for row in SLT.linklist]
    for seed, sigma, buckets, n, maxcycles in ((12345678901234567, 0.05, 8, 128, 30), (2 ** 63 - 25, 1e-9, 4, 64, 3),
                                               (977, 0.02, 16, 256, 40)):
        want, state = sample_linked_distances(OA, OB, ll, seed, sigma, buckets, n, maxcycles)
        got = SLT.sample_linked_distances(sigma=sigma, buckets=buckets, n=n, maxcycles=maxcycles, seed=seed)
        assert SLT._seed == state
        if want is None:
            assert got is None      # did not converge within maxcycles (pyx:3071)
            continue
        assert list(got) == ["TreeA", "TreeB", "n_pairs", "n_samples", "deviation_a", "deviation_b"]
        assert_bits_equal(got["TreeA"], want["TreeA"])
        assert_bits_equal(got["TreeB"], want["TreeB"])
        for k in ("n_pairs", "n_samples", "deviation_a", "deviation_b"):
            assert got[k] == want[k], k
        assert got["n_samples"] == len(got["TreeA"]) and got["n_samples"] % (buckets * n) == 0
        assert got["deviation_a"] < sigma and got["deviation_b"] < sigma
    # without seed= the sequence continues from where the object's generator is (pyx:2572: the state lives in the object)
    state = SLT._seed
    want, after = sample_linked_distances(OA, OB, ll, state, 0.05, 8, 128, 30)
    got = SLT.sample_linked_distances(sigma=0.05, buckets=8, n=128, maxcycles=30)
    assert SLT._seed == after and (want is None) == (got is None)
    if want is not None:
        assert_bits_equal(got["TreeA"], want["TreeA"])
    # the sampled distances estimate the exhaustive ones (what the method is for)
    full = SLT.linked_distances()
    est = SLT.sample_linked_distances(sigma=0.01, buckets=16, n=1024, seed=5)
    assert est is not None and abs(est["TreeA"].mean() - np.concatenate((full["TreeA"], full["TreeA"], np.zeros(SLT.n_links))).mean()) < 0.05 * full["TreeA"].mean() + 0.05


def test_quartet_topologies(ml_arrays):
    parent, dist, leaf_ids = ml_arrays
    T = SuchTree((parent, dist))
    O = OracleTree(parent, dist)
    rng = np.random.default_rng(8)
    q = rng.choice(leaf_ids, size=(60_000, 4))
    got = T.quartet_topologies_bulk(q)
    assert got.dtype == np.int64 and got.shape == q.shape
    assert np.array_equal(got, O.quartets(q))
    for strategy in ("walk", "canopy"):            # MRCA ids from either kernel family
        T._device_tree().set_strategy(strategy)
        assert np.array_equal(T.quartet_topologies_bulk(q), got), strategy
        assert np.array_equal(T.quartet_topologies_bulk(q[:100]), got[:100]), strategy
    assert np.array_equal(np.sort(got, axis=1), np.sort(q, axis=1))
    qi = rng.integers(0, len(parent), (20_000, 4))             # internal nodes, repeated ids
    assert np.array_equal(T.quartet_topologies_bulk(qi), O.quartets(qi))
    assert np.array_equal(T.quartet_topologies_bulk(np.asfortranarray(q[:5000])), O.quartets(q[:5000]))
    # several chunks through the three slots of the host pipe (524,288 quartets each), an id out of
    # range in a late chunk, and MRCA ids from the rank-table kernel or from the distance kernels
    big = rng.choice(leaf_ids, size=(1_700_001, 4))
    for ranks in (1, 0):
        T._device_tree().set_option("mrca_ranks", ranks)
        got_big = T.quartet_topologies_bulk(big)
        for lo in (0, 524_000, 1_048_500, 1_690_000):
            assert np.array_equal(got_big[lo:lo + 10_001], O.quartets(big[lo:lo + 10_001])), (ranks, lo)
    assert np.array_equal(np.sort(got_big, axis=1), np.sort(big, axis=1))
    bad = big.copy()
    bad[1_600_000, 2] = len(parent) + 11
    with pytest.raises(InvalidNodeError) as err:
        T.quartet_topologies_bulk(bad)
    assert err.value.node_id == len(parent) + 11
    assert np.array_equal(T.quartet_topologies_bulk(q), got)          # the pipe is clean after the error
    with pytest.raises(ValueError, match=r"Expected \(n, 4\) array"):
        T.quartet_topologies_bulk(np.zeros((3, 3), dtype=np.int64))
    with pytest.raises(InvalidNodeError):
        T.quartet_topologies_bulk(np.array([[0, 2, 4, len(parent)]]))
    G = SuchTree(golden_path("test.tree"))
    topo = G.quartet_topology("Oche", "Ocav", "Ohet", "Ound")
    assert topo == frozenset((frozenset(("Oche", "Ohet")), frozenset(("Ocav", "Ound"))))
    assert G.quartet_topology(0, 4, 2, 6) == frozenset((frozenset((0, 2)), frozenset((4, 6))))
    by_name = G.quartet_topologies_by_name([("Oche", "Ocav", "Ohet", "Ound"), ("Ttal", "Oche", "Tbot", "Ohet")])
    assert by_name[0] == topo
    assert by_name[1] == frozenset((frozenset(("Ttal", "Tbot")), frozenset(("Oche", "Ohet"))))


def test_config4_full_size_index_arithmetic():
    """The full 100k-leaf triangle has 4,999,950,000 pairs: pair indices beyond 2^32 must map to
    the right (row, column).  Slices at the far end and across the 2^31 / 2^32 boundaries."""
    m = 100_000
    parent, dist = synth.complete_tree(m, seed=44)
    dev = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    ids = np.arange(0, 2 * m, 2, dtype=np.int64)
    total = m * (m - 1) // 2
    assert total == 4_999_950_000
    for k0 in (total - 150_000, 2**31 - 70_000, 2**32 - 70_000, 0):
        c = 150_000 if k0 + 150_000 <= total else total - k0
        d, mm = dev.triangle_host(ids, k_begin=k0, k_count=c, want_dist=True, want_mrca=True)
        kk = np.arange(k0, k0 + c)
        rows = sharding.triangle_row_of(kk)
        cols = kk - rows * (rows - 1) // 2
        assert rows.max() <= m - 1 and cols.min() >= 0 and np.all(cols < rows)
        pp = np.stack([ids[cols], ids[rows]], 1)
        assert_bits_equal(d, O.distances(pp), "k0=%d" % k0)
        assert np.array_equal(mm, O.mrca_bulk(pp))
    with pytest.raises(ValueError):
        dev.triangle_host(ids, k_begin=total - 10, k_count=11)
    dev.close()


def test_more_than_2_31_pairs_in_one_launch():
    """Pair indices are 64-bit end to end: one device-resident launch over 2^31 + 2^20 pairs."""
    import torch
    parent, dist = synth.balanced_tree(16)
    dev = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    n = 2**31 + 2**20
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    base = torch.randint(0, 1 << 16, (1 << 22, 2), generator=g, device="cuda", dtype=torch.int64) * 2
    pairs = base.repeat(n // (1 << 22) + 1, 1)[:n].contiguous()        # 34 GB of ids in HBM
    out_d = torch.empty(n, dtype=torch.float64, device="cuda")
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    dev.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream)
    dev.fault_check(stream)
    host_base = base.cpu().numpy()
    want_d, want_m = O.distances(host_base), O.mrca_bulk(host_base)
    for start in (0, 2**31 - (1 << 21), n - (1 << 22)):
        start -= start % (1 << 22)
        sl = slice(start, start + (1 << 22))
        assert_bits_equal(out_d[sl].cpu().numpy(), want_d, "slice at %d" % start)
        assert np.array_equal(out_m[sl].cpu().numpy(), want_m)
    tail = out_d[n - (n % (1 << 22)):].cpu().numpy()
    assert_bits_equal(tail, want_d[: len(tail)])
    assert float(out_d.sum().item()) > 0
    dev.close()


def test_torch_interop_device_resident(ml_arrays):
    import torch
    from suchtree_amd import torch_interop
    parent, dist, leaf_ids = ml_arrays
    T = SuchTree((parent, dist))
    O = OracleTree(parent, dist)
    host = np.random.default_rng(21).choice(leaf_ids, size=(300_000, 2))
    pairs = torch.from_numpy(host).cuda()
    d, m = torch_interop.distances_device(T, pairs)
    assert d.dtype == torch.float64 and m.dtype == torch.int32 and d.is_cuda
    assert_bits_equal(d.cpu().numpy(), O.distances(host))
    assert np.array_equal(m.cpu().numpy(), O.mrca_bulk(host))
    # strided views: swapped columns, every other row
    d2, _ = torch_interop.distances_device(T, pairs[::2], want_mrca=False)
    assert_bits_equal(d2.cpu().numpy(), O.distances(host[::2]))
    cols = torch.from_numpy(np.ascontiguousarray(host.T)).cuda().t()      # column-major storage
    d3, m3 = torch_interop.distances_device(T, cols)
    assert_bits_equal(d3.cpu().numpy(), O.distances(host))
    d32, _ = torch_interop.distances_device(T, pairs, want_mrca=False, dist_dtype=torch.float32)
    assert d32.dtype == torch.float32 and torch.equal(d32.double(), d)
    # a C-order view whose base is only 8-byte aligned (no 16-byte vector loads possible)
    flat = torch.empty(2 * len(host) + 1, dtype=torch.int64, device="cuda")
    flat[1:] = pairs.reshape(-1)
    odd = flat[1:].view(-1, 2)
    assert odd.data_ptr() % 16 == 8
    d4, m4 = torch_interop.distances_device(T, odd)
    assert torch.equal(d4, d) and torch.equal(m4, m)
    bad = pairs.clone()
    bad[17, 1] = len(parent)
    with pytest.raises(InvalidNodeError):
        torch_interop.distances_device(T, bad)
    with pytest.raises(ValueError):
        torch_interop.distances_device(T, pairs.to(torch.int32))
    ids = torch.from_numpy(np.ascontiguousarray(leaf_ids[:500])).cuda()
    tri, _ = torch_interop.triangle_device(T, ids)
    rows, cols_ = np.tril_indices(500, -1)
    want = O.distances(np.stack([leaf_ids[:500][cols_], leaf_ids[:500][rows]], 1))
    assert_bits_equal(tri.cpu().numpy(), want)


def test_sharded_device_path_with_rccl_single_rank(ml_arrays):
    """The device-resident sharded call through a real RCCL process group (one rank here;
    the slicing / gather arithmetic for more ranks is covered by the gloo tests)."""
    import os
    import torch
    import torch.distributed as dist
    parent, dist_arr, leaf_ids = ml_arrays
    T = SuchTree((parent, dist_arr))
    O = OracleTree(parent, dist_arr)
    host = np.random.default_rng(31).choice(leaf_ids, size=(100_001, 2))
    pairs = torch.from_numpy(host).cuda()
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        d, m = sharding.distances_sharded_device(T, pairs)
        ds, ms, (lo, hi) = sharding.distances_sharded_device(T, pairs, gather=False)
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
    finally:
        dist.destroy_process_group()
    assert (lo, hi) == (0, 100_001) and torch.equal(ds, d) and torch.equal(ms, m)
    assert_bits_equal(d.cpu().numpy(), O.distances(host))
    assert np.array_equal(m.cpu().numpy(), O.mrca_bulk(host))


def test_wire_format_of_the_gather_on_one_gpu(ml_arrays):
    """What a peer's kernels write for the multi-GPU gather (st_distances_device_wire: float32 + 24-bit ids, packed by
    the kernels themselves) is what the root unpacks (st_unpack_mrca24_device, and the torch form the gloo tests
    use), piece by piece at 4-pair boundaries of a slice, for every kernel family."""
    import torch
    parent, dist_arr, leaf_ids = ml_arrays
    rng = np.random.default_rng(77)
    trees = [("ml.tree", parent, dist_arr, {}), ("ml.tree, walk family", parent, dist_arr, {"strategy": "walk"}),
             ("2^16 leaves", *synth.balanced_tree(16), {})]
    for what, par, dst, opts in trees:
        dev = _capi.DeviceTree(par, dst)
        for name, v in opts.items():
            dev.set_strategy(v) if name == "strategy" else dev.set_option(name, v)
        n = 700_003
        host = rng.integers(0, len(par), (n, 2))
        host[::11, 1] = host[::11, 0]
        pairs = torch.from_numpy(host).cuda()
        ref_d = torch.empty(n, dtype=torch.float32, device="cuda")
        ref_m = torch.empty(n, dtype=torch.int32, device="cuda")
        dev.distances_device(pairs.data_ptr(), n, ref_d.data_ptr(), ref_m.data_ptr(), f32=True)
        plan = sharding.ShardPlan(n, 1, 0, chunks=5, align=4)
        wire_d = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
        wire_m = torch.full((sharding.packed_bytes(n) + 16,), 0xEE, dtype=torch.uint8, device="cuda")
        for lo, hi in plan.pieces(0):
            assert lo % 4 == 0
            dev.distances_device_wire(pairs.data_ptr() + 16 * lo, hi - lo, wire_d.data_ptr() + 4 * lo, wire_m.data_ptr() + 3 * lo)
        dev.fault_check()
        assert torch.equal(wire_d.view(torch.int32), ref_d.view(torch.int32)), what
        assert bool((wire_m[sharding.packed_bytes(n):] == 0xEE).all()), what      # nothing written past the last dword
        got = torch.empty(n, dtype=torch.int32, device="cuda")
        for lo, hi in plan.pieces(0):      # the library's kernel, at any byte offset
            dev.unpack_mrca24_device(wire_m.data_ptr() + 3 * lo, hi - lo, got.data_ptr() + 4 * lo)
        assert torch.equal(got, ref_m), what
        got.fill_(-9)
        sharding.unpack_mrca24(wire_m, got)
        assert torch.equal(got, ref_m), what
        # ids only / distances only
        wire_m.fill_(0xEE)
        dev.distances_device_wire(pairs.data_ptr(), n, 0, wire_m.data_ptr())
        sharding.unpack_mrca24(wire_m, got)
        assert torch.equal(got, ref_m), what
        wire_d.fill_(-7.0)
        dev.distances_device_wire(pairs.data_ptr(), n, wire_d.data_ptr(), 0)
        assert torch.equal(wire_d.view(torch.int32), ref_d.view(torch.int32)), what
        # an id out of range travels as 0xFFFFFF and comes back as -1
        bad = pairs[:1000].clone()
        bad[5, 0] = len(par)
        dev.distances_device_wire(bad.data_ptr(), 1000, wire_d.data_ptr(), wire_m.data_ptr())
        sharding.unpack_mrca24(wire_m, got[:1000])
        assert int(got[5]) == -1 and torch.equal(got[6:1000], ref_m[6:1000])
        with pytest.raises(InvalidNodeError):
            dev.fault_check()
        with pytest.raises(Exception):
            dev.distances_device_wire(pairs.data_ptr(), 16, wire_d.data_ptr(), wire_m.data_ptr() + 1)      # not 4-byte aligned
        dev.close()


def _sum_of_all_pairwise_distances(parent, dist, n_leaves_total):
    """Closed form, float64: every edge e contributes length(e) * s(e) * (m - s(e)) to the sum over all
    unordered leaf pairs, s(e) = leaves below e.  Children have smaller depth-order than parents in a
    pass from the deepest node up; here a plain accumulation over nodes sorted by depth."""
    from suchtree_amd.newick import node_depths
    n = len(parent)
    depth = node_depths(parent)
    below = np.zeros(n, dtype=np.int64)
    is_leaf = np.ones(n, dtype=bool)
    is_leaf[parent[parent >= 0]] = False
    below[is_leaf] = 1
    for x in np.argsort(-depth, kind="stable"):          # deepest first: a node is complete before its parent
        p = parent[x]
        if p >= 0:
            below[p] += below[x]
    s = below.astype(np.float64)
    w = dist.astype(np.float64)
    inner = parent >= 0
    return float(np.sum(w[inner] * s[inner] * (n_leaves_total - s[inner])))


def test_config4_full_triangle_streamed_to_host(monkeypatch):
    """BASELINE config 4 at FULL size: all 4,999,950,000 pairs of the 100,000-leaf lower triangle in
    linked_distances order (MuchTree.pyx:2918-2925) streamed through st_triangle_host, tile by tile, into
    one reused host buffer.  Per tile: a sampled oracle check; on some tiles walk == canopy bit for bit;
    over the whole stream: exact pair count, the closed-form sum of all pairwise distances, and the same
    stream once more through a two-device handle (both entries device 0) so that the dealing of chunks
    over devices runs at this size -- every tile's checksum must be identical."""
    m = 100_000
    parent, dist = synth.complete_tree(m, seed=44)
    O = OracleTree(parent, dist)
    ids = np.arange(0, 2 * m, 2, dtype=np.int64)
    total = m * (m - 1) // 2
    assert total == 4_999_950_000
    tile = 1 << 26
    buf = np.empty(tile, dtype=np.float64)
    buf_w = np.empty(1 << 22, dtype=np.float64)
    rng = np.random.default_rng(44)

    def stream(dev, check):
        sums, done, t = [], 0, 0
        while done < total:
            c = min(tile, total - done)
            dev.triangle_host(ids, k_begin=done, k_count=c, out_dist=buf[:c])
            sums.append(float(buf[:c].sum()))
            if check:
                pick = np.unique(np.concatenate([[0, c - 1], rng.integers(0, c, 1500)]))
                kk = done + pick
                rows = sharding.triangle_row_of(kk)
                cols = kk - rows * (rows - 1) // 2
                assert_bits_equal(buf[pick], O.distances(np.stack([ids[cols], ids[rows]], 1)), "tile %d" % t)
                if t % 16 == 5:        # a window of the same tile by the other kernel family
                    w0 = int(rng.integers(0, max(1, c - len(buf_w))))
                    wn = min(len(buf_w), c - w0)
                    dev.set_strategy("walk")
                    dev.triangle_host(ids, k_begin=done + w0, k_count=wn, out_dist=buf_w[:wn])
                    dev.set_strategy("auto")
                    assert_bits_equal(buf_w[:wn], buf[w0:w0 + wn], "walk vs canopy, tile %d" % t)
            done += c
            t += 1
        assert done == total and t == (total + tile - 1) // tile
        return sums

    dev = _capi.DeviceTree(parent, dist)
    assert dev.info()["strategy"] == "canopy"
    sums = stream(dev, True)
    dev.close()
    want = _sum_of_all_pairwise_distances(parent, dist, m)
    assert abs(sum(sums) - want) <= 1e-6 * want          # float32 sums per pair, 5e9 of them, against float64

    monkeypatch.setenv("SUCHTREE_AMD_ALLOW_DUPLICATE_DEVICES", "1")
    T = SuchTree((parent, dist), devices=[0, 0])
    dev2 = T._device_tree()
    assert dev2.info()["n_devices"] == 2
    sums2 = stream(dev2, False)
    assert sums2 == sums
