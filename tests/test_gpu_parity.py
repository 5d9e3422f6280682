"""Parity of the HIP path (through the C ABI) with the oracle -- needs an MI355X.

Bar (BASELINE.json north_star): MRCA ids bit-exact; distances within 1e-6
relative.  Because the kernels reproduce the reference's float32 summation
order the tests demand more: the float64 outputs must be bit-identical.
"""
import os
import threading

import numpy as np
import pytest

from conftest import assert_bits_equal, golden_path, oracle_both
from oracle.oracle import OracleTree
from suchtree_amd import InvalidNodeError, SuchTree, _capi, synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6   # the stated tolerance; bit-equality below is stricter


def _both(dev, pairs):
    """Both kernel families with the settings the tree defaults to (the non-default canopy
    variants have one sweep test of their own, test_non_default_canopy_variants)."""
    out = {}
    for strategy in ("walk", "canopy"):
        try:
            dev.set_strategy(strategy)
        except ValueError:
            continue
        out[strategy] = dev.distances_host(pairs, want_dist=True, want_mrca=True)
    dev.set_strategy("auto")
    return out


def _check(parent, dist, pairs, strategy="auto"):
    want_d, want_m = oracle_both(parent, dist, pairs)
    dev = _capi.DeviceTree(parent, dist, strategy=strategy)
    res = _both(dev, pairs)
    assert res, "no kernel family ran"
    for name, (d, m) in res.items():
        assert np.array_equal(m, want_m), "%s mrca" % name
        fin = np.isfinite(want_d)       # sums that overflow to inf are covered by the bit test below
        assert np.all(np.abs(d[fin] - want_d[fin]) <= REL_TOL * np.abs(want_d[fin])), "%s outside 1e-6" % name
        assert_bits_equal(d, want_d, "%s distances" % name)
    d_only, none = dev.distances_host(pairs, want_dist=True, want_mrca=False)
    assert none is None
    assert_bits_equal(d_only, want_d)
    none, m_only = dev.distances_host(pairs, want_dist=False, want_mrca=True)
    assert none is None and np.array_equal(m_only, want_m)
    info = dev.info()
    dev.close()
    return info


def test_gopher_every_pair_against_golden():
    z = np.load(golden_path("gopher_all_pairs.npz"))
    info = _check(z["parent"], z["distance"], z["pairs"])
    dev = _capi.DeviceTree(z["parent"], z["distance"])
    for name, (d, m) in _both(dev, z["pairs"]).items():
        assert_bits_equal(d, z["dist"], name)
        assert np.array_equal(m, z["mrca"]), name
    big = np.tile(z["pairs"], (8, 1))           # > kCanopyMinPairs: the canopy kernels really run
    for name, (d, m) in _both(dev, big).items():
        assert_bits_equal(d, np.tile(z["dist"], 8), name)
        assert np.array_equal(m, np.tile(z["mrca"], 8)), name
    assert info["depth"] == 9 and info["n_nodes"] == 29


def test_config1_gopher_1000_random_leaf_pairs(gopher_flat):
    leaf_ids = np.array(list(gopher_flat.leaves.values()))
    pairs = np.random.default_rng(1).choice(leaf_ids, size=(1000, 2)).astype(np.int64)
    _check(gopher_flat.parent, gopher_flat.distance, pairs)
    # small batches: the pinned-mailbox path and the staged pipe must agree, for every batch
    # size around the mailbox limit and for strided views
    O = OracleTree(gopher_flat.parent, gopher_flat.distance)
    dev = _capi.DeviceTree(gopher_flat.parent, gopher_flat.distance)
    big = np.random.default_rng(2).choice(leaf_ids, size=(5000, 2)).astype(np.int64)
    for n in (1, 2, 63, 64, 65, 1000, 2047, 2048, 2049, 5000, 8191, 8192, 8193):
        for mailbox in (1, 0):
            dev.set_option("small_batch_path", mailbox)
            for view in (big[:n], np.asfortranarray(big[:n]), big[:n][:, ::-1]):
                d, m = dev.distances_host(view, True, True)
                assert_bits_equal(d, O.distances(view), "n=%d mailbox=%d" % (n, mailbox))
                assert np.array_equal(m, O.mrca_bulk(view))
            d, _ = dev.distances_host(big[:n], True, False)
            assert_bits_equal(d, O.distances(big[:n]))
            _, m = dev.distances_host(big[:n], False, True)
            assert np.array_equal(m, O.mrca_bulk(big[:n]))
    dev.close()


@pytest.mark.parametrize("which", ["ml", "nj"])
def test_config2_bigtrees(which, ml_arrays, nj_arrays):
    parent, dist, leaf_ids = ml_arrays if which == "ml" else nj_arrays
    rng = np.random.default_rng(2)
    info = _check(parent, dist, rng.choice(leaf_ids, size=(300_000, 2)))
    # deep canopies (hundreds of levels) get the smaller LDS image and longer understory chains
    assert info["strategy"] == "canopy" and info["canopy_nodes"] <= 10240
    assert info["record_bytes"] == (128 if which == "ml" else 256)
    _check(parent, dist, rng.integers(0, len(parent), (100_000, 2)))      # internal nodes too
    a = np.arange(0, 60_000)
    _check(parent, dist, np.stack([a, a + rng.integers(0, 7, a.size)], 1))   # shared portals / understory MRCAs


@pytest.mark.parametrize("levels", [3, 10, 14, 17])
def test_balanced_trees(levels):
    parent, dist = synth.balanced_tree(levels)
    n = len(parent)
    rng = np.random.default_rng(levels)
    _check(parent, dist, rng.integers(0, n, (200_000, 2)))
    a = np.arange(0, min(n - 9, 100_000))
    _check(parent, dist, np.stack([a, a + rng.integers(0, 9, a.size)], 1))


def test_config3_balanced_2_20_sample_and_full_size_properties():
    levels, n_leaves = 20, 1 << 20
    parent, dist = synth.balanced_tree(levels)
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert (info["strategy"], info["canopy_nodes"], info["understory_max"], info["record_bytes"]) == \
        ("canopy", 16383, 7, 64)
    # oracle on a sample the CPU finishes in seconds
    O = OracleTree(parent, dist)
    sample = synth.random_leaf_pairs(n_leaves, 1_000_000, seed=3)
    want_d, want_m = O.distances(sample), O.mrca_bulk(sample)
    for name, (d, m) in _both(dev, sample).items():
        assert_bits_equal(d, want_d, name)
        assert np.array_equal(m, want_m), name
    # a fifth of the bench step's batch through the host path, both families: size-independent properties
    # (the whole 1e8-pair batch, device resident, exactly as bench.py launches it: test_config3_headline_launch_1e8_device_resident)
    big = synth.random_leaf_pairs(n_leaves, 20_000_000, seed=11)
    res = _both(dev, big)
    (dw, mw), (dc, mc) = res["walk"], res["canopy"]
    for name, (d, m) in res.items():
        assert_bits_equal(d, dw, name + " vs walk at 2e7 pairs")
        assert np.array_equal(m, mw), name
    assert_bits_equal(dc, dw, "canopy vs walk at 2e7 pairs")
    assert np.array_equal(mc, mw)
    # mrca(a,b) == mrca(b,a); |d(a,b) - d(b,a)| within float32 rounding of the sum
    d_rev, m_rev = dev.distances_host(big[:, ::-1], True, True)     # negative-stride view
    assert np.array_equal(m_rev, mc)
    assert np.all(np.abs(d_rev - dc) <= 4e-6 * dc + 1e-12)
    # the MRCA of two leaves of a complete tree: leaves differ first at bit k of the leaf index
    la, lb = big[:, 0] >> 1, big[:, 1] >> 1
    same = la == lb
    x = la ^ lb
    k = np.zeros(len(x), dtype=np.int64)
    nz = ~same
    k[nz] = np.floor(np.log2(x[nz].astype(np.float64))).astype(np.int64) + 1
    expect = np.where(same, big[:, 0], (((la >> k) << 1 | 1) << k) - 1)
    assert np.array_equal(mc.astype(np.int64), expect)
    assert np.all(dc[same] == 0.0)
    dev.close()


def test_config3_headline_launch_1e8_device_resident():
    """bench.py's timed call itself (bench.py::HipBackend.bind -> st_distances_device): SURVEY 8d config 3's batch --
    default_rng(3).integers(0, 2^20, (1e8, 2)) * 2, int64 ids resident in HBM -- in ONE launch into float64 + int32
    device buffers, i.e. k_canopy_ilp<7, 1, SrcContig, true> with the plain sinks.  MRCA ids: the closed form of a complete
    tree on all 1e8 pairs; distances and ids: the oracle's bits on every 100th pair and on the first and last 1e5."""
    import torch
    levels, n_leaves, n = 20, 1 << 20, 100_000_000
    parent, dist = synth.balanced_tree(levels)
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert (info["strategy"], info["record_bytes"], info["a_side_bytes"]) == ("canopy", 64, 4)      # the headline kernel's tables
    host = synth.random_leaf_pairs(n_leaves, n, seed=3)
    pairs = torch.from_numpy(host).cuda()
    out_d = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
    out_m = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    dev.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    dev.fault_check(stream)
    # closed form, all pairs: two leaves of a complete tree differ first at bit k of the leaf index
    la, lb = pairs[:, 0] >> 1, pairs[:, 1] >> 1
    x = la ^ lb
    k = torch.where(x > 0, torch.floor(torch.log2(x.clamp(min=1).double())).long() + 1, torch.zeros_like(x))
    expect = torch.where(x == 0, pairs[:, 0], (((la >> k) << 1 | 1) << k) - 1)
    assert torch.equal(out_m.long(), expect)
    assert bool((out_d[x == 0] == 0.0).all()) and bool((out_d[x != 0] > 0.0).all())
    del la, lb, x, k, expect
    # oracle bits: a strided sample of 1e6 pairs and both ends of the batch
    O = OracleTree(parent, dist)
    cores = len(os.sched_getaffinity(0))
    idx = np.unique(np.concatenate([np.arange(0, n, 100), np.arange(0, 100_000), np.arange(n - 100_000, n)]))
    it = torch.from_numpy(idx).cuda()
    got_d, got_m = out_d[it].cpu().numpy(), out_m[it].cpu().numpy()
    assert_bits_equal(got_d, O.distances_mt(host[idx], cores), "headline launch, sampled distances")
    assert np.array_equal(got_m, O.mrca_bulk(host[idx]))
    dev.close()


def test_deep_and_random_and_tiny_trees():
    rng = np.random.default_rng(9)
    parent, dist = synth.caterpillar_tree(3000)
    _check(parent, dist, rng.integers(0, len(parent), (20_000, 2)))
    parent, dist = synth.random_binary_tree(50_000, seed=1, zero_fraction=0.1)
    _check(parent, dist, rng.integers(0, len(parent), (200_000, 2)))
    parent, dist = synth.random_binary_tree(1, seed=1)
    _check(parent, dist, np.array([[0, 0]]))
    parent, dist = synth.random_binary_tree(2, seed=1)
    _check(parent, dist, np.array([[0, 0], [0, 2], [2, 0], [1, 2], [0, 1], [1, 1]]))


def test_tree_too_deep_for_the_canopy_falls_back_to_walk():
    parent, dist = synth.caterpillar_tree(40_000)
    with pytest.raises(_capi.TreeStructureError):
        _capi.DeviceTree(parent, dist, strategy="canopy")
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, len(parent), (3000, 2))
    info = _check(parent, dist, pairs, strategy="auto")
    assert info["strategy"] == "walk" and info["depth"] == 40_000
    # mean depth 20,000: lineage tables would take 13 GB for a 2 MB tree -- capped at 2048 entries per node, it climbs
    assert info["lineage_entries"] == 0 and info["device_bytes"] < 64 << 20


def test_walk_only_tree_with_sparse_table_and_lineage_sums():
    """A large deep tree the canopy family refuses (120,000 leaves, depth ~840): the walk family
    takes the meeting node from the whole-tree sparse table and a's side from lineage sums addressed
    by node id; both switched off as well.  Leaf and internal nodes, near pairs, (x, x)."""
    import torch
    rng = np.random.default_rng(7)
    parent, dist = _random_shape_tree(rng, 120_000, 0.97)
    n = len(parent)
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert info["strategy"] == "walk" and info["lineage_entries"] > n, info
    O = OracleTree(parent, dist)
    pairs = rng.integers(0, n, (12_000, 2))
    a = rng.integers(0, n - 30, 4_000)
    allp = np.concatenate([pairs, np.stack([a, a + rng.integers(0, 30, a.size)], 1), np.stack([a[:500], a[:500]], 1)]).astype(np.int64)
    want_d, want_m = O.distances(allp), O.mrca_bulk(allp)
    t = torch.from_numpy(allp).cuda()
    for rmq, sums in ((1, 1), (0, 1), (1, 0), (0, 0)):
        dev.set_option("tree_rmq", rmq)
        dev.set_option("lineage_sums", sums)
        out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
        out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
        dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
        dev.fault_check()
        assert_bits_equal(out_d.cpu().numpy(), want_d, "rmq=%d sums=%d" % (rmq, sums))
        assert np.array_equal(out_m.cpu().numpy(), want_m)
        d, m = dev.distances_host(allp[:3000], True, True)            # mailbox
        assert_bits_equal(d, want_d[:3000], "mailbox rmq=%d sums=%d" % (rmq, sums))
        assert np.array_equal(m, want_m[:3000])
        assert np.array_equal(dev.distances_host(allp, False, True)[1], want_m)
    dev.close()


@pytest.mark.parametrize("which", ["walk_only", "ml"])
def test_tile_sorted_walk_kernel_and_its_tables(which, ml_arrays):
    """k_walk_sorted (batches >= 262144 pairs on trees with the sparse table and both lineage tables) and
    every table it builds on, switched on and off in all combinations: tile sort, crown (shared portal
    blocks + crown sparse table), the crown's ladder in LDS, lineage lengths, whole-tree sparse table.  A walk-only tree (the canopy
    family refuses it) and ml.tree with the walk family forced.  Leaves and internal nodes, near pairs,
    (x, x), a batch that is no multiple of the tile, device buffers and the host path."""
    import itertools
    import torch
    rng = np.random.default_rng(17)
    if which == "walk_only":
        parent, dist = _random_shape_tree(rng, 120_000, 0.97)
        dev = _capi.DeviceTree(parent, dist)
        assert dev.info()["strategy"] == "walk"
    else:
        parent, dist, _ = ml_arrays
        dev = _capi.DeviceTree(parent, dist)
        dev.set_strategy("walk")
    n = len(parent)
    assert dev.info()["lineage_entries"] > n
    O = OracleTree(parent, dist)
    a = rng.integers(0, n - 30, 20_000)
    allp = np.concatenate([rng.integers(0, n, (540_001, 2)), np.stack([a, a + rng.integers(0, 30, a.size)], 1),
                           np.stack([a[:2000], a[:2000]], 1)]).astype(np.int64)
    cores = len(os.sched_getaffinity(0))
    want_d, want_m = O.distances_mt(allp, cores), O.mrca_bulk(allp)
    t = torch.from_numpy(allp).cuda()
    out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
    out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
    for srt, crown, lens, rmq, lad in itertools.product((1, 0), repeat=5):
        if lad and not (srt and crown and lens):
            continue                      # the ladder form only exists inside the sorted kernel with crown and lengths on
        for name, v in (("walk_sort", srt), ("walk_crown", crown), ("lineage_lens", lens), ("tree_rmq", rmq), ("walk_ladder", lad)):
            dev.set_option(name, v)
        what = "sort=%d crown=%d lens=%d rmq=%d ladder=%d" % (srt, crown, lens, rmq, lad)
        out_d.fill_(-1.0)
        dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
        dev.fault_check()
        assert_bits_equal(out_d.cpu().numpy(), want_d, what)
        assert np.array_equal(out_m.cpu().numpy(), want_m), what
    for name in ("walk_sort", "walk_crown", "lineage_lens", "tree_rmq", "walk_ladder"):
        dev.set_option(name, 1)
    # every tile size (the default picks one by batch size), with and without the ladder; a batch just above the
    # kernel's smallest
    for tile, lad in itertools.product((1, 2, 4), (1, 0)):
        dev.set_option("sort_tile", tile)
        dev.set_option("walk_ladder", lad)
        for n_dev in (len(allp), 262_144 + 77):
            out_d.fill_(-1.0)
            dev.distances_device(t.data_ptr(), n_dev, out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d[:n_dev].cpu().numpy(), want_d[:n_dev], "sort_tile=%d ladder=%d n=%d" % (tile, lad, n_dev))
            assert np.array_equal(out_m[:n_dev].cpu().numpy(), want_m[:n_dev])
            assert out_d[n_dev:n_dev + 64].eq(-1.0).all() or n_dev == len(allp)
    dev.set_option("sort_tile", 0)
    dev.set_option("walk_ladder", 1)
    # distances only / MRCA ids only, float32 sink, the host path (pinned staging read once, coalesced stores)
    dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), 0)
    assert_bits_equal(out_d.cpu().numpy(), want_d, "distances only")
    d, m = dev.distances_host(allp, True, True)
    assert_bits_equal(d, want_d, "host path")
    assert np.array_equal(m, want_m)
    assert_bits_equal(dev.distances_host(allp.astype(np.int32), True, False)[0], want_d, "host path, int32 ids")
    # an id out of range inside a sorted tile: reported like the reference reports it, the handle stays usable
    bad = allp.copy()
    bad[40_000, 1] = n + 5
    with pytest.raises(InvalidNodeError) as err:
        dev.distances_host(bad, True, True)
    assert err.value.node_id == n + 5
    tb = torch.from_numpy(bad).cuda()
    dev.distances_device(tb.data_ptr(), len(bad), out_d.data_ptr(), out_m.data_ptr())
    with pytest.raises(InvalidNodeError):
        dev.fault_check()
    got = out_d.cpu().numpy()
    assert np.isnan(got[40_000]) and int(out_m[40_000].item()) == -1
    keep = np.arange(len(bad)) != 40_000
    assert_bits_equal(got[keep], want_d[keep], "the rest of a batch with a bad id")
    dev.close()


def test_walk_tables_are_optional(monkeypatch):
    """SUCHTREE_AMD_WALK_TABLE_MB=0: a tree only the walk family serves is built without the sparse table
    and the lineage tables (they are aids, never requirements) and answers by climbing."""
    rng = np.random.default_rng(3)
    parent, dist = _random_shape_tree(rng, 30_000, 0.97)
    pairs = rng.integers(0, len(parent), (50_000, 2))
    O = OracleTree(parent, dist)
    want_d, want_m = O.distances(pairs), O.mrca_bulk(pairs)
    for mb, has_tables in (("0", False), ("1", False), ("4096", True)):
        monkeypatch.setenv("SUCHTREE_AMD_WALK_TABLE_MB", mb)
        dev = _capi.DeviceTree(parent, dist, strategy="walk")
        info = dev.info()
        assert (info["lineage_entries"] > 0) == has_tables, (mb, info)
        d, m = dev.distances_host(pairs, True, True)
        assert_bits_equal(d, want_d, "budget %s MB" % mb)
        assert np.array_equal(m, want_m)
        dev.close()


def test_four_byte_a_side_of_the_predicated_kernel():
    """rec_a4: on balanced-like trees the first node of a pair costs a 4-byte gather (its understory sum;
    the portal comes from a block table in LDS).  On and off, leaves (the fast path) and internal nodes
    (the 8-byte fallback), device buffers, int32 host path; a random tree, where the form is not built."""
    import torch
    for parent, dist in (synth.balanced_tree(16), synth.complete_tree(50_000, seed=2), synth.random_binary_tree(40_000, seed=9)):
        n = len(parent)
        rng = np.random.default_rng(n)
        dev = _capi.DeviceTree(parent, dist)
        O = OracleTree(parent, dist)
        leaf = np.arange(0, n, 2)
        pairs = np.concatenate([rng.choice(leaf, size=(150_000, 2)), rng.integers(0, n, (50_000, 2))]).astype(np.int64)
        want_d, want_m = O.distances_mt(pairs, len(os.sched_getaffinity(0))), O.mrca_bulk(pairs)
        t = torch.from_numpy(pairs).cuda()
        out_d = torch.empty(len(pairs), dtype=torch.float64, device="cuda")
        out_m = torch.empty(len(pairs), dtype=torch.int32, device="cuda")
        for on, cherries in ((1, 1), (1, 0), (0, 1)):      # (cherries: b's record shared by two sibling leaves, where they exist)
            dev.set_option("rec_a4", on)
            dev.set_option("cherries", cherries)
            dev.distances_device(t.data_ptr(), len(pairs), out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d.cpu().numpy(), want_d, "rec_a4=%d cherries=%d" % (on, cherries))
            assert np.array_equal(out_m.cpu().numpy(), want_m)
            d, m = dev.distances_host(pairs.astype(np.int32), True, True)
            assert_bits_equal(d, want_d, "rec_a4=%d host int32" % on)
            assert np.array_equal(m, want_m)
        dev.close()
    # where the cherry records are built: the balanced tree (every leaf) and the complete tree (its two lowest levels mix)
    for (parent, dist), expect in ((synth.balanced_tree(16), 8), (synth.complete_tree(50_000, seed=2), None), (synth.random_binary_tree(40_000, seed=9), 32)):
        info = _capi.DeviceTree(parent, dist).info()
        if expect is not None:
            assert info["b_table_bytes_per_leaf"] == expect, info      # (record_bytes / 4 with cherries, / 2 without)


def test_special_float_values():
    parent, dist = synth.random_binary_tree(3000, seed=4)
    rng = np.random.default_rng(4)
    dist = dist.copy()
    k = rng.integers(0, len(dist), 600)
    dist[k[:100]] = np.float32(1e-42)          # denormal: must not be flushed
    dist[k[100:200]] = np.float32(-0.0)
    dist[k[200:300]] = np.float32(3e38)        # overflows to inf in long sums
    dist[k[300:400]] = np.float32(-1.5)        # NJ trees carry negative lengths
    dist[k[400:500]] = np.float32(2.220446e-16)
    dist[k[500:]] = np.float32(1.17549435e-38)
    _check(parent, dist, rng.integers(0, len(parent), (100_000, 2)))


def test_strides_and_dtypes_through_the_facade(ml_arrays):
    parent, dist, leaf_ids = ml_arrays
    T = SuchTree((parent, dist))
    O = OracleTree(parent, dist)
    pairs = np.random.default_rng(7).choice(leaf_ids, size=(50_000, 2))
    want = O.distances(pairs)
    assert_bits_equal(T.distances_bulk(pairs), want)
    assert_bits_equal(T.distances_bulk(np.asfortranarray(pairs)), want)
    wide = np.zeros((50_000, 6), dtype=np.int64)
    wide[:, 1::3] = pairs
    assert_bits_equal(T.distances_bulk(wide[:, 1::3]), want)
    assert_bits_equal(T.distances_bulk(pairs[::-1])[::-1], want)
    p32 = pairs.astype(np.int32)
    assert_bits_equal(T.distances_bulk(p32), want)                                   # int32 entry point
    assert_bits_equal(T.distances_bulk(np.asfortranarray(p32)), want)
    assert_bits_equal(T.distances_bulk(p32[:1500]), want[:1500])                     # mailbox path, int32
    assert_bits_equal(T.distances_bulk(pairs.astype(np.uint16 if len(parent) < 65536 else np.uint32)), want)
    big32 = np.random.default_rng(8).choice(leaf_ids, size=(5_000_000, 2)).astype(np.int32)
    pick32 = np.random.default_rng(9).integers(0, len(big32), 100_000)
    assert_bits_equal(T.distances_bulk(big32)[pick32], O.distances(big32[pick32].astype(np.int64)))
    bad32 = p32.copy()
    bad32[5, 0] = -4
    with pytest.raises(InvalidNodeError) as e32:
        T.distances_bulk(bad32)
    assert e32.value.node_id == -4
    assert_bits_equal(T.distances_bulk(pairs[:100].tolist()), want[:100])
    # more than one pipeline chunk (2^22 pairs) of a strided view
    big = np.asfortranarray(np.random.default_rng(8).choice(leaf_ids, size=(6_000_000, 2)))
    got = T.distances_bulk(big)
    pick = np.random.default_rng(9).integers(0, len(big), 200_000)
    assert_bits_equal(got[pick], O.distances(np.ascontiguousarray(big[pick])))
    assert_bits_equal(got[-1000:], O.distances(np.ascontiguousarray(big[-1000:])))
    d, m = T.distances_and_ancestors_bulk(pairs)
    assert_bits_equal(d, want)
    assert np.array_equal(m, O.mrca_bulk(pairs))
    assert np.array_equal(T.common_ancestors_bulk(pairs), m)


def test_out_of_range_ids_raise_like_the_reference(gopher_flat):
    T = SuchTree(golden_path("test.tree"))
    n = T.size
    cases = [
        (np.array([[0, 2], [n, 4]]), n),                 # max too large -> max reported
        (np.array([[0, 2], [-3, 4]]), -3),               # only negative -> min reported
        (np.array([[-7, n + 5], [1, 2]]), n + 5),        # both -> max reported (MuchTree.pyx:897-903)
        (np.array([[0, 2**40]]), 2**40),                 # wider than the int32 transport of the host path
        (np.array([[0, 2**40], [2**41 + 5, 1]]), 2**41 + 5),
        (np.array([[-2**40, 2], [-5, 1]]), -2**40),
        (np.array([[-2**40, n]]), n),
        (np.array([[2**31 - 1, 0]]), 2**31 - 1),
        (np.array([[-2**31, 0]]), -2**31),
    ]
    for pairs, bad in cases:
        with pytest.raises(InvalidNodeError) as e:
            T.distances_bulk(pairs)
        assert e.value.node_id == bad and e.value.tree_size == n
        assert str(e.value) == "Node ID %d out of bounds (tree size: %d)" % (bad, n)
    # the handle is still usable and the fault word was cleared
    assert T.distance(0, 2) == OracleTree(gopher_flat.parent, gopher_flat.distance).distance(0, 2)
    big = np.random.default_rng(0).integers(0, n, (10_000, 2))
    big[7777, 1] = n
    with pytest.raises(InvalidNodeError):
        T.distances_bulk(big)                             # canopy-sized batch


def test_device_resident_buffers_and_fault_word(ml_arrays):
    import ctypes
    parent, dist, leaf_ids = ml_arrays
    L = _capi.load()
    dev = _capi.DeviceTree(parent, dist)
    O = OracleTree(parent, dist)
    n = 400_000
    pairs = np.random.default_rng(3).choice(leaf_ids, size=(n, 2))
    ptrs = []
    for nbytes in (n * 16, n * 8, n * 4):
        p = ctypes.c_void_p()
        assert L.st_device_malloc(0, nbytes, ctypes.byref(p)) == 0
        ptrs.append(p)
    d_pairs, d_dist, d_mrca = ptrs
    assert L.st_memcpy_h2d(0, d_pairs, pairs.ctypes.data_as(ctypes.c_void_p), n * 16) == 0
    for strategy in ("walk", "canopy"):
        dev.set_strategy(strategy)
        dev.distances_device(d_pairs.value, n, d_dist.value, d_mrca.value)
        dev.fault_check()
        out_d = np.zeros(n)
        out_m = np.zeros(n, dtype=np.int32)
        assert L.st_memcpy_d2h(0, out_d.ctypes.data_as(ctypes.c_void_p), d_dist, n * 8) == 0
        assert L.st_memcpy_d2h(0, out_m.ctypes.data_as(ctypes.c_void_p), d_mrca, n * 4) == 0
        assert_bits_equal(out_d, O.distances(pairs), strategy)
        assert np.array_equal(out_m, O.mrca_bulk(pairs))
    # column-major device layout: stride0 = 1, stride1 = n
    cols = np.ascontiguousarray(pairs.T)
    assert L.st_memcpy_h2d(0, d_pairs, cols.ctypes.data_as(ctypes.c_void_p), n * 16) == 0
    dev.distances_device(d_pairs.value, n, d_dist.value, 0, stride0=1, stride1=n)
    dev.fault_check()
    out_d = np.zeros(n)
    assert L.st_memcpy_d2h(0, out_d.ctypes.data_as(ctypes.c_void_p), d_dist, n * 8) == 0
    assert_bits_equal(out_d, O.distances(pairs))
    # a bad id is reported by the fault word, never dereferenced
    bad = pairs.copy()
    bad[123, 0] = len(parent) + 9
    assert L.st_memcpy_h2d(0, d_pairs, bad.ctypes.data_as(ctypes.c_void_p), n * 16) == 0
    dev.distances_device(d_pairs.value, n, d_dist.value, d_mrca.value)
    with pytest.raises(InvalidNodeError) as e:
        dev.fault_check()
    assert e.value.node_id == len(parent) + 9
    dev.fault_check()   # cleared
    for p in ptrs:
        assert L.st_device_free(0, p) == 0
    dev.close()


def test_concurrent_callers_share_one_handle(ml_arrays):
    parent, dist, leaf_ids = ml_arrays
    T = SuchTree((parent, dist)).to_device()
    O = OracleTree(parent, dist)
    rng = np.random.default_rng(5)
    batches = [rng.choice(leaf_ids, size=(60_000, 2)) for _ in range(4)]
    want = [O.distances(b) for b in batches]
    got = [None] * 4

    def work(i):
        got[i] = T.distances_bulk(batches[i])

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for g, w in zip(got, want):
        assert_bits_equal(g, w)


def _random_shape_tree(rng, n_leaves, skew):
    return synth.skewed_tree(rng, n_leaves, skew)


def test_record_sizes_and_shapes_sweep():
    """Every record stride the library can choose (16 ... 512 bytes) and both climb regimes,
    on trees from perfectly balanced to nearly caterpillar."""
    rng = np.random.default_rng(77)
    seen = set()
    cases = [(lv, None) for lv in range(11, 19)]                       # balanced: R = 16, 32, 64
    cases += [(int(rng.integers(2000, 60000)), float(s)) for s in (0.0, 0.1, 0.3, 0.5, 0.7, 0.85, 0.95, 0.99)]
    for size, skew in cases:
        if skew is None:
            parent, dist = synth.balanced_tree(size)
        else:
            parent, dist = _random_shape_tree(rng, size, skew)
        n = len(parent)
        pairs = rng.integers(0, n, (60_000, 2))
        a = rng.integers(0, n - 40, 30_000)
        near = np.stack([a, a + rng.integers(0, 40, a.size)], 1)
        info = _check(parent, dist, np.concatenate([pairs, near]))
        seen.add((info["strategy"], info["record_bytes"]))
    strides = {r for s, r in seen if s == "canopy"}
    # (128-byte records: ml.tree in test_config2_bigtrees)
    assert {16, 32, 64, 256, 512}.issubset(strides), seen


def test_lineage_sum_mode_of_the_deep_kernel(ml_arrays):
    """Deep canopies with in-order ids: the tile-sorted kernel reads a's whole side of a pair from
    the lineage-sum table.  Deep random shapes and ml.tree; leaf and internal nodes, pairs under a
    shared portal, (x, x), ancestor / descendant pairs, generated triangle, out-of-range ids;
    the table on and off."""
    import torch
    rng = np.random.default_rng(404)
    # (since round 6 the tile-sorted kernel serves chains of at most seven slots -- small deep trees: 16-, 32- and 64-byte records
    # below; on longer records the same options run the scalar ladder kernel, whose joint form reads the same table)
    trees = [_random_shape_tree(rng, 30000, 0.97), _random_shape_tree(rng, 9000, 0.995), _random_shape_tree(rng, 11000, 0.9),
             _random_shape_tree(rng, 16000, 0.9), (ml_arrays[0], ml_arrays[1])]
    sorted_records = set()
    for parent, dist in trees:
        n = len(parent)
        O = OracleTree(parent, dist)
        dev = _capi.DeviceTree(parent, dist)
        info = dev.info()
        assert info["strategy"] == "canopy" and info["lineage_entries"] > n, info
        dev.set_option("tile_sort", 1)
        dev.set_option("ladder_scalar", 0)
        dev.set_option("prefer_walk_sorted", 0)
        if dev.info()["big_batch_kernel"] == "canopy_sorted":
            sorted_records.add(info["record_bytes"])
        elif info["record_bytes"] >= 128:
            dev.set_option("ladder_scalar", 1)
            assert dev.info()["big_batch_kernel"] == "canopy_ladder", dev.info()
        # (else: short records under a canopy image that leaves the tile-sorted kernel no scratch -- the predicated kernel runs)
        pairs = rng.integers(0, n, (200_000, 2))
        a = rng.integers(0, n - 12, 50_000)
        near = np.stack([a, a + rng.integers(0, 12, a.size)], 1)           # mostly one portal
        up = rng.integers(0, n, 20_000)
        anc = up.copy()
        for _ in range(int(rng.integers(1, 60))):                           # an ancestor of `up`
            anc = np.where(parent[anc] >= 0, parent[anc], anc)
        lineage = np.concatenate([np.stack([up, anc], 1), np.stack([anc, up], 1), np.stack([up, up], 1)])
        allp = np.concatenate([pairs, near, lineage]).astype(np.int64)
        want_d, want_m = oracle_both(parent, dist, allp)
        ids = rng.choice(n, size=400, replace=False).astype(np.int64)
        i, j = np.tril_indices(len(ids), -1)
        tri = np.stack([ids[j], ids[i]], 1)
        tri_d, tri_m = oracle_both(parent, dist, tri)
        t = torch.from_numpy(allp).cuda()
        for on in (1, 0, 1):
            dev.set_option("lineage_sums", on)
            out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
            out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
            dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d.cpu().numpy(), want_d, "lineage_sums=%d" % on)
            assert np.array_equal(out_m.cpu().numpy(), want_m)
            for n_host in (70_001, 140_001):       # walk kernel on pinned memory / staged tile-sorted kernel
                d, m = dev.distances_host(allp[:n_host], True, True)
                assert_bits_equal(d, want_d[:n_host], "host n=%d lineage_sums=%d" % (n_host, on))
                assert np.array_equal(m, want_m[:n_host])
            m_only = dev.distances_host(allp[:9_000], False, True)[1]
            assert np.array_equal(m_only, want_m[:9_000])
            # every tile size the tile-sorted kernel is built with (the default picks one by batch size)
            for tile in (1, 2, 4, 0):
                dev.set_option("sort_tile", tile)
                out_d.fill_(-5.0)
                out_m.fill_(-5)
                dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
                dev.fault_check()
                assert_bits_equal(out_d.cpu().numpy(), want_d, "lineage_sums=%d sort_tile=%d" % (on, tile))
                assert np.array_equal(out_m.cpu().numpy(), want_m)
            if on:      # either side of the walk / tile-sorted switch: 131072 pairs, in HBM and from the host
                for n_dev in (131071, 131072, 131073):
                    out_d.fill_(-5.0)
                    dev.distances_device(t.data_ptr(), n_dev, out_d.data_ptr(), out_m.data_ptr())
                    dev.fault_check()
                    assert_bits_equal(out_d[:n_dev].cpu().numpy(), want_d[:n_dev], "device n=%d" % n_dev)
                    assert np.array_equal(out_m[:n_dev].cpu().numpy(), want_m[:n_dev])
                    assert out_d[n_dev:n_dev + 64].eq(-5.0).all()
                for n_host in (131071, 131072, 131073):
                    d, m = dev.distances_host(allp[:n_host], True, True)
                    assert_bits_equal(d, want_d[:n_host], "host n=%d" % n_host)
                    assert np.array_equal(m, want_m[:n_host])
            # MRCA ids only through the tile-sorted kernel (they are complete after its key phase)
            out_m.fill_(-5)
            dev.distances_device(t.data_ptr(), len(allp), 0, out_m.data_ptr())
            dev.fault_check()
            assert np.array_equal(out_m.cpu().numpy(), want_m)
            # distances only
            out_d.fill_(-5.0)
            dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), 0)
            dev.fault_check()
            assert_bits_equal(out_d.cpu().numpy(), want_d, "distances only, lineage_sums=%d" % on)
            td, tm = dev.triangle_host(ids, want_dist=True, want_mrca=True)
            assert_bits_equal(td, tri_d, "triangle lineage_sums=%d" % on)
            assert np.array_equal(tm, tri_m)
            bad = allp[:150_000].copy()
            bad[131_337, 0] = n + 5
            with pytest.raises(_capi.InvalidNodeError) as err:
                dev.distances_host(bad, True, True)
            assert err.value.node_id == n + 5
        dev.close()
    # (16-byte records at least: 32- / 64-byte records mostly come with canopies of ~10,000 nodes, whose image leaves the tile no scratch)
    assert sorted_records and max(sorted_records) <= 64, sorted_records


def test_every_candidate_kernel_on_nj_tree(nj_arrays):
    """nj.tree (256-byte records: 31-slot chains, the shared-portal case of the predicated kernel in two steps):
    every candidate kernel forced by options, uniform and nearby pairs, both outputs and MRCA ids alone."""
    import torch
    parent, dist, leaf_ids = nj_arrays
    n = len(parent)
    rng = np.random.default_rng(31)
    a = rng.integers(0, n - 40, 150_000)
    allp = np.concatenate([rng.integers(0, n, (450_000, 2)), np.stack([a, a + rng.integers(0, 40, a.size)], 1)]).astype(np.int64)
    want_d, want_m = oracle_both(parent, dist, allp)
    t = torch.from_numpy(allp).cuda()
    out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
    out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
    dev = _capi.DeviceTree(parent, dist)
    assert dev.info()["record_bytes"] == 256
    seen = set()
    for sort, walk, ladder in ((1, 0, 0), (0, 0, 0), (1, 1, 0), (0, 0, 1)):
        dev.set_option("tile_sort", sort)
        dev.set_option("prefer_walk_sorted", walk)
        dev.set_option("ladder_scalar", ladder)
        kernel = dev.info()["big_batch_kernel"]
        seen.add(kernel)
        out_d.fill_(-1.0)
        out_m.fill_(-1)
        dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
        dev.fault_check()
        assert_bits_equal(out_d.cpu().numpy(), want_d, kernel)
        assert np.array_equal(out_m.cpu().numpy(), want_m), kernel
        d, m = dev.distances_host(allp, True, True)      # (packed ids on the way back)
        assert_bits_equal(d, want_d, "host path, " + kernel)
        assert np.array_equal(m, want_m), kernel
    assert seen == {"canopy", "walk_sorted", "canopy_ladder"}, seen      # (tile_sort = 1 selects nothing on 31-slot chains since round 6: the predicated kernel runs)
    out_m.fill_(-1)
    dev.distances_device(t.data_ptr(), len(allp), 0, out_m.data_ptr())      # k_mrca_ranks<31>
    dev.fault_check()
    assert np.array_equal(out_m.cpu().numpy(), want_m)
    dev.close()


@pytest.mark.parametrize("which", ["ml", "nj", "cap63", "cap127"])
def test_ladder_kernel_joint_form_keeps_the_bits(which, ml_arrays, nj_arrays):
    """The scalar ladder kernel's joint form (option ladder_sums: a's whole side from the lineage sums, the meeting node from
    rec_p + the 64-bit sparse table, b's record by chunks, one LDS climb per pair) against the oracle and against the form
    that climbs both sides: 15-, 31-, 63-slot chains in registers and 127-slot chains through a pointer; every kind of
    node (leaves, internal nodes, canopy nodes, the root), equal nodes, ancestor / descendant pairs, neighbours under one
    portal; static and dynamic deal; device-resident and through the host path's wire formats."""
    import torch
    rng = np.random.default_rng(61)
    if which == "ml":
        parent, dist = ml_arrays[0], ml_arrays[1]
    elif which == "nj":
        parent, dist = nj_arrays[0], nj_arrays[1]
    elif which == "cap63":
        parent, dist = _random_shape_tree(rng, 54_000, 0.95)
    else:
        parent, dist = _random_shape_tree(np.random.default_rng(5), 1_000_000, 0.9)      # (bench.py's walk_only_tree leg: 1 KB records)
    n = len(parent)
    scale = 5 if which == "cap127" else 1      # (the oracle's visited-list scan is O(depth^2) per pair: fewer pairs on the deepest tree)
    pairs = rng.integers(0, n, (700_000 // scale, 2))
    a = rng.integers(0, n - 70, 60_000 // scale)
    near = np.stack([a, a + rng.integers(0, 70, a.size)], 1)
    same = np.stack([a[:3000], a[:3000]], 1)
    root = int(np.flatnonzero(parent < 0)[0])
    up = a[:20_000 // scale].copy()
    for _ in range(3):      # a node and its great-grandparent (the root where the lineage is shorter)
        up = np.where(parent[up] >= 0, parent[up], root)
    anc = np.stack([a[:up.size], up], 1)
    with_root = np.stack([np.full(2000, root), a[:2000]], 1)
    allp = np.concatenate([pairs, near, same, anc, anc[:, ::-1], with_root, with_root[:, ::-1]]).astype(np.int64)
    want_d, want_m = oracle_both(parent, dist, allp)
    dev = _capi.DeviceTree(parent, dist, strategy="canopy")
    info = dev.info()
    assert info["record_bytes"] == {"ml": 128, "nj": 256, "cap63": 512}.get(which, info["record_bytes"])
    if which == "cap127" and info["record_bytes"] != 1024:
        pytest.skip("this shape did not need 1 KB records (record_bytes %d)" % info["record_bytes"])
    t = torch.from_numpy(allp).cuda()
    out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
    out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
    for k, v in (("tile_sort", 0), ("ladder_scalar", 1), ("ladder_min_pairs", 0), ("prefer_walk_sorted", 0), ("batch_probe", 0)):
        dev.set_option(k, v)
    assert dev.info()["big_batch_kernel"] == "canopy_ladder"
    for sums in (1, 0):
        for dynamic in (0, 1):
            dev.set_option("ladder_sums", sums)
            dev.set_option("ladder_dynamic", dynamic)
            if which != "cap127":      # (the 1e6-leaf tree's lineage sums exceed the canopy family's limit: the option is a wish there, the climbing form runs)
                assert dev.info()["ladder_sums"] == sums
            out_d.fill_(-1.0)
            out_m.fill_(-7)
            dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d.cpu().numpy(), want_d, "%s sums=%d dynamic=%d" % (which, sums, dynamic))
            assert np.array_equal(out_m.cpu().numpy(), want_m), (which, sums, dynamic)
        d, m = dev.distances_host(allp, True, True)      # (float32 + 24-bit ids over the link)
        assert_bits_equal(d, want_d, "%s host path sums=%d" % (which, sums))
        assert np.array_equal(m, want_m), (which, sums)
        d, _ = dev.distances_host(allp, True, False)
        assert_bits_equal(d, want_d, "%s host path, distances alone, sums=%d" % (which, sums))
    # without the lineage sums the option is a wish the handle cannot grant: the climbing form runs, the info says so
    dev.set_option("lineage_sums", 0)
    dev.set_option("ladder_sums", 1)
    assert dev.info()["ladder_sums"] == 0
    dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
    dev.fault_check()
    assert_bits_equal(out_d.cpu().numpy(), want_d, which + " no lineage sums")
    dev.close()


@pytest.mark.parametrize("which", ["gopher", "ml", "nj"])
def test_gpu_against_the_committed_vectors_of_the_reference_s_compiled_hot_path(which, gopher_flat, ml_arrays, nj_arrays):
    """tests/golden/ref_hotpath_vectors.npz (outputs of the reference's own compiled _distances / _mrca / _quartet_topologies on
    seeded inputs, scripts/make_ref_golden.py): the HIP path reproduces them bit for bit -- data, no library needed."""
    z = np.load(golden_path("ref_hotpath_vectors.npz"))
    parent, dist = {"gopher": (gopher_flat.parent, gopher_flat.distance), "ml": ml_arrays[:2], "nj": nj_arrays[:2]}[which]
    pairs = z[which + "_pairs"].astype(np.int64)
    want_d, want_m = z[which + "_dist"].astype(np.float64), z[which + "_mrca"]
    dev = _capi.DeviceTree(parent, dist)
    for name, (d, m) in _both(dev, pairs).items():
        assert_bits_equal(d, want_d, which + " " + name)
        assert np.array_equal(m, want_m), (which, name)
    dev.close()
    from suchtree_amd import SuchTree
    T = SuchTree((np.asarray(parent), np.asarray(dist)), device=0)
    assert np.array_equal(np.asarray(T.quartet_topologies_bulk(z[which + "_quartets"].astype(np.int64))), z[which + "_topologies"].astype(np.int64))


@pytest.mark.parametrize("which", ["ml", "nj", "balanced17", "deep63"])
def test_gpu_against_the_reference_s_own_compiled_hot_path(which, ml_arrays, nj_arrays):
    """The HIP path against the REFERENCE's compiled SuchTree._distances / _mrca (oracle/_ref/libref_hotpath.so: MuchTree.c as shipped,
    compiled where it lies, oracle/ref_harness.c -- the file travels with the snapshot), not through the oracle's restatement:
    distances bit for bit (float32 ordered sums widened to float64), MRCA ids exactly; device-resident pairs through the C ABI and
    numpy pairs through the host path; every kind of node pair."""
    import torch
    from oracle import oracle as orc
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref/libref_hotpath.so did not travel with this snapshot")
    rng = np.random.default_rng(97)
    if which == "ml":
        parent, dist = ml_arrays[0], ml_arrays[1]
    elif which == "nj":
        parent, dist = nj_arrays[0], nj_arrays[1]
    elif which == "balanced17":
        parent, dist = synth.balanced_tree(17)
    else:
        parent, dist = _random_shape_tree(rng, 54_000, 0.95)
    n = len(parent)
    root = int(np.flatnonzero(parent < 0)[0])
    a = rng.integers(0, n, 60_000)
    up = a.copy()
    for _ in range(int(rng.integers(1, 30))):
        up = np.where(parent[up] >= 0, parent[up], root)
    allp = np.concatenate([rng.integers(0, n, (500_000, 2)), np.stack([a, np.clip(a + rng.integers(-30, 31, a.size), 0, n - 1)], 1),
                           np.stack([a[:5000], a[:5000]], 1), np.stack([a, up], 1), np.stack([up, a], 1)]).astype(np.int64)
    R = orc.RefTree(parent, dist)
    cores = len(os.sched_getaffinity(0))
    want_d, want_m = R.distances(allp, cores), R.mrca_bulk(allp[:250_000])
    dev = _capi.DeviceTree(parent, dist)
    t = torch.from_numpy(allp).cuda()
    out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
    out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
    dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
    dev.fault_check()
    assert_bits_equal(out_d.cpu().numpy(), want_d, which + ": device path vs the reference's code")
    assert np.array_equal(out_m[:250_000].cpu().numpy(), want_m), which
    d, m = dev.distances_host(allp, True, True)
    assert_bits_equal(d, want_d, which + ": host path vs the reference's code")
    assert np.array_equal(m[:250_000], want_m), which
    dev.set_strategy("walk")
    d, m = dev.distances_host(allp[:300_000], True, True)
    assert_bits_equal(d, want_d[:300_000], which + ": walk family vs the reference's code")
    assert np.array_equal(m[:250_000], want_m), which
    dev.close()


def test_kernel_of_large_batches_is_timed_at_creation(monkeypatch, tmp_path):
    """Deep trees: the handle times its candidate kernels when it is created and makes the fastest its default
    (st_tree_info.tuned / big_batch_kernel); every candidate, forced by options, gives the same bits.  512-byte
    records (63-slot chains in registers in the predicated kernel), a batch large enough for every kernel."""
    import torch
    rng = np.random.default_rng(5)
    parent, dist = _random_shape_tree(rng, 54_000, 0.95)
    n = len(parent)
    O = OracleTree(parent, dist)
    pairs = rng.integers(0, n, (600_000, 2))
    a = rng.integers(0, n - 70, 40_000)
    allp = np.concatenate([pairs, np.stack([a, a + rng.integers(0, 70, a.size)], 1), np.stack([a[:2000], a[:2000]], 1)]).astype(np.int64)
    cores = len(os.sched_getaffinity(0))
    want_d, want_m = O.distances_mt(allp, cores), O.mrca_bulk(allp)
    t = torch.from_numpy(allp).cuda()
    out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
    out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")

    def run(dev, what):
        out_d.fill_(-1.0)
        out_m.fill_(-1)
        dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
        dev.fault_check()
        assert_bits_equal(out_d.cpu().numpy(), want_d, what)
        assert np.array_equal(out_m.cpu().numpy(), want_m), what

    # the decision is recorded per (tree, device, library build): the first handle times (tuned = 1), later ones --
    # in this or any other process -- read the record (tuned = 2) and run the same kernel
    monkeypatch.setenv("SUCHTREE_AMD_CACHE_DIR", str(tmp_path / "tune"))
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert info["strategy"] == "canopy" and info["record_bytes"] == 512 and info["lineage_entries"] > 0, info
    assert info["tuned"] == 1 and info["big_batch_kernel"] in ("canopy", "walk_sorted", "canopy_ladder"), info
    assert len(list((tmp_path / "tune").glob("tune-*.txt"))) == 1
    again = _capi.DeviceTree(parent, dist)
    assert again.info()["tuned"] == 2 and again.info()["big_batch_kernel"] == info["big_batch_kernel"], again.info()
    again.close()
    monkeypatch.setenv("SUCHTREE_AMD_TUNE_CACHE", "0")
    timed = _capi.DeviceTree(parent, dist)
    assert timed.info()["tuned"] == 1
    timed.close()
    monkeypatch.delenv("SUCHTREE_AMD_TUNE_CACHE")
    run(dev, "default (%s)" % info["big_batch_kernel"])
    seen = set()
    for sort, walk, ladder in ((1, 0, 0), (0, 0, 0), (1, 1, 0), (0, 0, 1)):
        dev.set_option("tile_sort", sort)
        dev.set_option("prefer_walk_sorted", walk)
        dev.set_option("ladder_scalar", ladder)
        kernel = dev.info()["big_batch_kernel"]
        seen.add(kernel)
        run(dev, kernel)
        d, m = dev.distances_host(allp[:300_000], True, True)
        assert_bits_equal(d, want_d[:300_000], "host path, " + kernel)
        assert np.array_equal(m, want_m[:300_000])
    assert seen == {"canopy", "walk_sorted", "canopy_ladder"}, seen
    dev.close()
    monkeypatch.setenv("SUCHTREE_AMD_AUTOTUNE", "0")
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert info["tuned"] == 0 and info["big_batch_kernel"] == "canopy_ladder" and info["ladder_sums"] == 0, info      # (the fixed rule: the ladder kernel where its image fits)
    run(dev, "rule (%s)" % info["big_batch_kernel"])
    dev.close()


def test_deep_canopy_tree_with_walk_form_lineage_tables(ml_arrays, monkeypatch):
    """A deep canopy tree whose lineage tables are too large for the canopy family's 28-bit offsets gets them in the
    walk family's form instead (offsets by node id; forced here on ml.tree by SUCHTREE_AMD_LINEAGE_MAX_ENTRIES): the
    walk kernels serve it as they serve trees without a canopy, the tile-sorted canopy kernel runs without lineage
    sums.  Every kernel against the oracle, device pairs and the host path."""
    import torch
    monkeypatch.setenv("SUCHTREE_AMD_LINEAGE_MAX_ENTRIES", "1000")
    parent, dist, leaf_ids = ml_arrays
    n = len(parent)
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert info["strategy"] == "canopy" and info["lineage_entries"] > n, info
    O = OracleTree(parent, dist)
    rng = np.random.default_rng(12)
    a = rng.integers(0, n - 30, 30_000)
    allp = np.concatenate([rng.integers(0, n, (560_000, 2)), np.stack([a, a + rng.integers(0, 30, a.size)], 1)]).astype(np.int64)
    cores = len(os.sched_getaffinity(0))
    want_d, want_m = O.distances_mt(allp, cores), O.mrca_bulk(allp)
    t = torch.from_numpy(allp).cuda()
    out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
    out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
    seen = set()
    for sort, walk in ((1, 0), (0, 0), (1, 1)):
        dev.set_option("tile_sort", sort)
        dev.set_option("prefer_walk_sorted", walk)
        dev.set_option("ladder_scalar", 0)      # (the handle may have chosen it; its own test: ..._is_timed_at_creation)
        kernel = dev.info()["big_batch_kernel"]
        seen.add(kernel)
        for n_dev in (len(allp), 140_000, 20_000, 3_000):      # sorted kernels, mid-sized batches, k_walk, mailbox size
            out_d.fill_(-1.0)
            dev.distances_device(t.data_ptr(), n_dev, out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d[:n_dev].cpu().numpy(), want_d[:n_dev], "%s n=%d" % (kernel, n_dev))
            assert np.array_equal(out_m[:n_dev].cpu().numpy(), want_m[:n_dev]), kernel
        d, m = dev.distances_host(allp[:300_000], True, True)
        assert_bits_equal(d, want_d[:300_000], "host path, " + kernel)
        assert np.array_equal(m, want_m[:300_000])
        d, m = dev.distances_host(allp[:2_000], True, True)
        assert_bits_equal(d, want_d[:2_000], "mailbox, " + kernel)
    assert seen == {"canopy", "walk_sorted"}, seen      # (15-slot chains: no tile-sorted canopy kernel since round 6)
    dev.set_strategy("walk")
    dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
    assert_bits_equal(out_d.cpu().numpy(), want_d, "walk family")
    dev.close()


def test_table_budget_on_the_device(ml_arrays):
    """A budget for the device tables (SuchTree(..., table_mb=) / st_tree_create_ex / SUCHTREE_AMD_TABLE_MB): tables are
    left out in the stated order, st_tree_info says which and stays within the budget, and every form that takes over
    gives the oracle's bits -- random pairs, pairs under one portal (walked on the tree once the id chains are gone),
    MRCA ids alone, the host path, generated pairs.  ml.tree down to 32 MiB and below; a 2^24-leaf tree of random shape
    (33.5 M nodes, 10 GB of tables without a budget) under 2 GiB."""
    import torch
    parent, dist, leaf_ids = ml_arrays
    n = len(parent)
    rng = np.random.default_rng(707)
    a = rng.integers(0, n - 12, 60_000)
    allp = np.concatenate([rng.integers(0, n, (200_000, 2)), np.stack([a, a + rng.integers(0, 12, a.size)], 1),
                           np.stack([a[:3000], a[:3000]], 1)]).astype(np.int64)
    want_d, want_m = oracle_both(parent, dist, allp)
    t = torch.from_numpy(allp).cuda()
    seen = set()
    for mb in (None, 100, 40, 32, 12, 6, 3.5):
        dev = _capi.DeviceTree(parent, dist, table_mb=mb)
        info = dev.info()
        seen.add(tuple(info["dropped_tables"]))
        if mb is not None:
            assert info["table_budget_bytes"] == int(mb * 2**20) and info["device_bytes"] <= mb * 2**20, info
        out_d = torch.empty(len(allp), dtype=torch.float64, device="cuda")
        out_m = torch.empty(len(allp), dtype=torch.int32, device="cuda")
        dev.distances_device(t.data_ptr(), len(allp), out_d.data_ptr(), out_m.data_ptr())
        dev.fault_check()
        assert_bits_equal(out_d.cpu().numpy(), want_d, "table_mb=%s %s" % (mb, info["dropped_tables"]))
        assert np.array_equal(out_m.cpu().numpy(), want_m), (mb, info["dropped_tables"])
        out_m.fill_(-5)
        dev.distances_device(t.data_ptr(), len(allp), 0, out_m.data_ptr())
        assert np.array_equal(out_m.cpu().numpy(), want_m), (mb, "MRCA ids alone")
        for n_host in (3_000, 50_000, len(allp)):
            d, m = dev.distances_host(allp[:n_host], True, True)
            assert_bits_equal(d, want_d[:n_host], "host path table_mb=%s n=%d" % (mb, n_host))
            assert np.array_equal(m, want_m[:n_host])
        dev.close()
    assert () in seen and any("rec_i" in s and "canopy" not in s for s in seen) and any("canopy" in s for s in seen), seen
    # the predicated kernel of a shallow canopy without its id chains: pairs of nearby leaves are walked
    par, dst = synth.balanced_tree(17)
    O = OracleTree(par, dst)
    a = rng.integers(0, len(par) - 9, 150_000)
    near = np.concatenate([np.stack([a, a + rng.integers(0, 9, a.size)], 1), rng.integers(0, len(par), (150_000, 2))]).astype(np.int64)
    dev = _capi.DeviceTree(par, dst, table_mb=20)
    info = dev.info()
    assert "rec_i" in info["dropped_tables"] and info["strategy"] == "canopy" and info["device_bytes"] <= 20 * 2**20, info
    d, m = dev.distances_host(near, True, True)
    assert_bits_equal(d, O.distances(near), "2^17 leaves without id chains")
    assert np.array_equal(m, O.mrca_bulk(near))
    assert np.array_equal(dev.distances_host(near, False, True)[1], O.mrca_bulk(near))
    dev.close()
    # 2^24 leaves of random shape under 2 GiB: the floor (940 MB) and whatever else fits; the walk kernel climbs
    par, dst = synth.random_binary_tree_levels(1 << 24, seed=24)
    dev = _capi.DeviceTree(par, dst, table_mb=2048)
    info = dev.info()
    assert info["strategy"] == "walk" and "canopy" in info["dropped_tables"] and info["device_bytes"] <= 2048 * 2**20, info
    pairs = rng.integers(0, len(par), (400_000, 2))
    d, m = dev.distances_host(pairs, True, True)
    O = OracleTree(par, dst)
    k = 60_000
    assert_bits_equal(d[:k], O.distances(pairs[:k]), "2^24 leaves under 2 GiB")
    assert np.array_equal(m[:k], O.mrca_bulk(pairs[:k]))
    dev.close()


def test_mrca_only_requests_from_the_rank_table(ml_arrays):
    """MRCA ids without distances on trees with in-order ids: k_mrca_ranks (rank of either portal
    + sparse table; shared-portal pairs through the understory records).  Shallow and deep tree,
    internal nodes, near pairs, (x, x), ancestor pairs, out-of-range ids; the option off as well."""
    import torch
    rng = np.random.default_rng(909)
    for parent, dist in (synth.balanced_tree(16), synth.random_binary_tree(40000, seed=5), (ml_arrays[0], ml_arrays[1])):
        n = len(parent)
        O = OracleTree(parent, dist)
        dev = _capi.DeviceTree(parent, dist)
        pairs = rng.integers(0, n, (150_000, 2))
        a = rng.integers(0, n - 9, 50_000)
        near = np.stack([a, a + rng.integers(0, 9, a.size)], 1)
        up = rng.integers(0, n, 10_000)
        anc = up.copy()
        for _ in range(5):
            anc = np.where(parent[anc] >= 0, parent[anc], anc)
        allp = np.concatenate([pairs, near, np.stack([up, anc], 1), np.stack([anc, up], 1), np.stack([up, up], 1)]).astype(np.int64)
        want = O.mrca_bulk(allp)
        t = torch.from_numpy(allp).cuda()
        for on in (1, 0):
            dev.set_option("mrca_ranks", on)
            out_m = torch.full((len(allp) + 8,), -7, dtype=torch.int32, device="cuda")
            dev.distances_device(t.data_ptr(), len(allp), 0, out_m.data_ptr())
            dev.fault_check()
            assert np.array_equal(out_m[:len(allp)].cpu().numpy(), want), on
            assert out_m[len(allp):].eq(-7).all()
            m = dev.distances_host(allp, False, True)[1]
            assert np.array_equal(m, want), on
            bad = allp[:30_000].copy()
            bad[12_345, 1] = -4
            with pytest.raises(_capi.InvalidNodeError) as err:
                dev.distances_host(bad, False, True)
            assert err.value.node_id == -4
        dev.close()


def test_general_trees_through_the_c_abi():
    """Arbitrary arity and arbitrary node numbering (not what the facade produces, but what the
    C ABI accepts): records in identity order, checked against the plain-Python restatement."""
    from oracle.oracle import py_distances, py_mrca
    from test_tables_emulated import _general_tree
    for n, max_children in ((400, 2), (3000, 3), (20000, 8)):
        rng = np.random.default_rng(n)
        parent, dist = _general_tree(rng, n, max_children)
        pairs = rng.integers(0, n, (6000, 2))
        want_d = py_distances(parent, dist, pairs)
        want_m = np.array([py_mrca(parent, int(a), int(b)) for a, b in pairs])
        dev = _capi.DeviceTree(parent, dist)
        for name, (d, m) in _both(dev, pairs).items():
            assert_bits_equal(d, want_d, name)
            assert np.array_equal(m, want_m), name
        # the tile-sorted ladder kernel on ids that are not in-order positions: no sparse table,
        # the meeting node comes from the lock-step search on the ladder
        dev.set_strategy("canopy")
        dev.set_option("tile_sort", 1)
        d, m = dev.distances_host(pairs, True, True)
        assert_bits_equal(d, want_d, "tile-sorted, lock-step")
        assert np.array_equal(m, want_m)
        dev.close()


def test_deep_tree_with_ids_that_are_not_in_order_positions(ml_arrays):
    """ml.tree with its node ids permuted (what the C ABI accepts, not what the facade produces): no sparse tables, so the
    scalar ladder kernel and the tile-sorted kernel find the meeting node by the lock-step search -- third ancestors from
    the ladder image (links are places since round 4), single steps from the 8-byte canopy entries in global memory."""
    parent, dist, leaf_ids = ml_arrays
    n = len(parent)
    rng = np.random.default_rng(11)
    new_id = rng.permutation(n).astype(np.int64)                 # old id -> new id
    p2 = np.empty(n, dtype=np.int32)
    d2 = np.empty(n, dtype=np.float32)
    p2[new_id] = np.where(parent >= 0, new_id[np.maximum(parent, 0)], -1).astype(np.int32)
    d2[new_id] = dist
    pairs = new_id[leaf_ids[rng.integers(0, len(leaf_ids), (300_000, 2))]]
    pairs[:1000, 1] = pairs[:1000, 0]                            # same node; and neighbours under one portal
    pairs[1000:3000] = new_id[leaf_ids[np.stack((np.arange(2000), np.arange(2000) + 1), axis=1)]]
    want_d, want_m = oracle_both(p2, d2, pairs)
    dev = _capi.DeviceTree(p2, d2, strategy="canopy")
    assert dev.info()["record_bytes"] >= 128
    for opts in ({"tile_sort": 0, "ladder_scalar": 1, "ladder_min_pairs": 0, "ladder_dynamic": 0},
                 {"tile_sort": 0, "ladder_scalar": 1, "ladder_min_pairs": 0, "ladder_dynamic": 1},
                 {"tile_sort": 1, "ladder_scalar": 0},
                 {"tile_sort": 0, "ladder_scalar": 0}):
        for k, v in opts.items():
            dev.set_option(k, v)
        d, m = dev.distances_host(pairs, True, True)
        assert_bits_equal(d, want_d, str(opts))
        assert np.array_equal(m, want_m), opts
    dev.close()


def test_batch_sizes_around_kernel_tile_boundaries(ml_arrays):
    """Batch lengths at and around every granule the launch code knows: the walk/canopy switch
    (4096 pairs), the 1024-pair workgroup tile, the 2048 / 4096-pair tiles of the tile-sorted
    kernel, a few workgroups more or less -- shallow and deep tree, device-resident pairs."""
    import torch
    rng = np.random.default_rng(77)
    sizes = [1, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6144, 6145,
             8191, 8192, 8193, 2048 * 512 - 1, 2048 * 512 + 1, 4096 * 256 + 4095]
    for parent, dist in (synth.balanced_tree(15), (ml_arrays[0], ml_arrays[1])):
        O = OracleTree(parent, dist)
        dev = _capi.DeviceTree(parent, dist)
        big = rng.integers(0, len(parent), (max(sizes), 2))
        big[::97, 1] = big[::97, 0]                                  # some pairs (x, x)
        want_d, want_m = O.distances(big), O.mrca_bulk(big)
        t = torch.from_numpy(big).cuda()
        for n in sizes:
            out_d = torch.full((n + 8,), -1.0, dtype=torch.float64, device="cuda")
            out_m = torch.full((n + 8,), -7, dtype=torch.int32, device="cuda")
            dev.distances_device(t.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d[:n].cpu().numpy(), want_d[:n], "n=%d" % n)
            assert np.array_equal(out_m[:n].cpu().numpy(), want_m[:n]), n
            assert out_d[n:].eq(-1.0).all() and out_m[n:].eq(-7).all(), "wrote past n=%d" % n
        dev.close()


@pytest.mark.parametrize("which", ["ml", "nj"])
def test_batch_probe_keeps_the_bits(which, ml_arrays, nj_arrays):
    """Large explicit batches on a deep tree are dealt by the batch probe (pair_math.h: probe_says_walk, run by every workgroup of the
    two kernels it chooses between): pairs of nearby leaves
    go to the tile-sorted walk kernel, uniform pairs to the scalar ladder kernel, a half-and-half batch to whichever the
    sample says -- the results are the oracle's bits every time, and the same with the probe switched off."""
    import torch
    parent, dist, leaf_ids = ml_arrays if which == "ml" else nj_arrays
    rng = np.random.default_rng(31)
    n = 4_400_000      # (the probe looks at batches of 2^22 pairs and more)
    ia = rng.integers(0, len(leaf_ids), n)
    near = np.stack([leaf_ids[ia], leaf_ids[np.clip(ia + rng.integers(-8, 9, n), 0, len(leaf_ids) - 1)]], 1).astype(np.int64)
    uniform = rng.choice(leaf_ids, size=(n, 2)).astype(np.int64)
    mixed = np.where((np.arange(n) % 2 == 0)[:, None], near, uniform)
    dev = _capi.DeviceTree(parent, dist)
    for k, v in (("tile_sort", 0), ("ladder_scalar", 1), ("ladder_min_pairs", 0), ("prefer_walk_sorted", 0)):
        dev.set_option(k, v)      # (the ladder kernel as the handle's choice, whatever it timed)
    out_d = torch.empty(n, dtype=torch.float64, device="cuda")
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    cores = len(os.sched_getaffinity(0))
    O = OracleTree(parent, dist)
    for name, batch in (("near", near), ("uniform", uniform), ("mixed", mixed)):
        want_d, want_m = O.distances_mt(batch, cores), O.mrca_bulk(batch[:200_000])
        t = torch.from_numpy(batch).cuda()
        for probe in (1, 0):
            dev.set_option("batch_probe", probe)
            out_d.fill_(-1.0)
            out_m.fill_(-7)
            for _ in range(3):      # (back to back: the probe's words are reused)
                dev.distances_device(t.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
            dev.fault_check()
            assert_bits_equal(out_d.cpu().numpy(), want_d, "%s probe=%d" % (name, probe))
            assert np.array_equal(out_m[:200_000].cpu().numpy(), want_m), name
    # which kernel a batch got: near -> walk (1), uniform -> ladder (0); and a batch of PERIODIC structure -- one pair in
    # eight is a near one, at the positions a fixed-stride sample of 4096 would land on -- is not mistaken for a near batch
    dev.set_option("batch_probe", 1)
    periodic = np.where((np.arange(n) % 8 == 0)[:, None], near, uniform)
    for name, batch, want in (("near", near, 1), ("uniform", uniform, 0), ("periodic", periodic, 0)):
        t = torch.from_numpy(batch).cuda()
        dev.distances_device(t.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
        assert dev.probe_last_choice() == want, name
    n2 = 1 << 22      # 1024 samples, one per 4096 pairs: at a FIXED stride every sample position would be one of the near pairs
    t = torch.from_numpy(np.ascontiguousarray(periodic[:n2])).cuda()
    dev.distances_device(t.data_ptr(), n2, out_d.data_ptr(), out_m.data_ptr())
    assert dev.probe_last_choice() == 0
    before = dev.probe_last_choice()
    dev.distances_device(t.data_ptr(), n2 - 1, out_d.data_ptr(), out_m.data_ptr())      # below the probe's smallest batch: the handle's kernel, no verdict
    assert dev.probe_last_choice() == before
    bad = near.copy()
    bad[::2048, 1] = len(parent) + 5      # ids out of range scattered through the batch: the probe skips them, the kernels report them
    t = torch.from_numpy(bad).cuda()
    dev.set_option("batch_probe", 1)
    dev.distances_device(t.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
    with pytest.raises(InvalidNodeError):
        dev.fault_check()
    dev.close()


def test_counter_and_probe_slots_survive_many_launches_on_several_streams():
    """The scalar ladder kernel's work counters are a ring of 64 slots on the handle; a slot's reuse is ordered behind its last
    user by an event, whatever stream that was on (the batch probe's words only report its verdict since round 6: every workgroup
    samples the batch itself).  200 launches of 2^22 pairs (the
    size at which 512-byte records draw their work dynamically) dealt round-robin over three streams without any host
    synchronisation in between: every launch must produce the whole, correct result."""
    import torch
    parent, dist = synth.skewed_tree(np.random.default_rng(5), 1_000_000, 0.8)
    dev = _capi.DeviceTree(parent, dist)
    info = dev.info()
    assert info["record_bytes"] == 512 and info["big_batch_kernel"] == "canopy_ladder", info
    n = 1 << 22
    pairs = torch.from_numpy(synth.random_leaf_pairs(1_000_000, n, seed=8)).cuda()
    ref_d = torch.empty(n, dtype=torch.float64, device="cuda")
    ref_m = torch.empty(n, dtype=torch.int32, device="cuda")
    dev.distances_device(pairs.data_ptr(), n, ref_d.data_ptr(), ref_m.data_ptr())
    torch.cuda.synchronize()
    dev.fault_check()
    O = OracleTree(parent, dist)
    k = 100_000
    assert_bits_equal(ref_d[:k].cpu().numpy(), O.distances_mt(pairs[:k].cpu().numpy(), len(os.sched_getaffinity(0))))
    assert np.array_equal(ref_m[:k].cpu().numpy(), O.mrca_bulk(pairs[:k].cpu().numpy()))
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [(torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")) for _ in streams]
    torch.cuda.synchronize()
    for launch in range(200):
        s = streams[launch % 3]
        d, m = outs[launch % 3]
        with torch.cuda.stream(s):
            d.fill_(-1.0)      # (on the launch's own stream: a pair a kernel skips keeps the sentinel)
            m.fill_(-7)
            dev.distances_device(pairs.data_ptr(), n, d.data_ptr(), m.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    for d, m in outs:
        assert torch.equal(d.view(torch.int64), ref_d.view(torch.int64)) and torch.equal(m, ref_m)
    dev.close()


def test_predicated_kernel_on_a_shallow_and_a_deep_tree(ml_arrays):
    """The predicated canopy kernel forced (no tile sort, no ladder) on a shallow and a deep tree, explicit pairs and the
    generated triangle."""
    rng = np.random.default_rng(21)
    trees = [synth.balanced_tree(16), (ml_arrays[0], ml_arrays[1])]
    for parent, dist in trees:
        O = OracleTree(parent, dist)
        dev = _capi.DeviceTree(parent, dist, strategy="canopy")
        pairs = rng.integers(0, len(parent), (150_000, 2))
        ids = rng.choice(len(parent), size=500, replace=False).astype(np.int64)
        i, j = np.tril_indices(len(ids), -1)
        tri = np.stack([ids[j], ids[i]], 1)
        for k, v in (("tile_sort", 0), ("ladder_scalar", 0), ("prefer_walk_sorted", 0)):
            dev.set_option(k, v)
        d, m = dev.distances_host(pairs, want_dist=True, want_mrca=True)
        td, tm = dev.triangle_host(ids, want_dist=True, want_mrca=True)
        assert_bits_equal(d, O.distances(pairs))
        assert_bits_equal(td, O.distances(tri), "triangle")
        assert np.array_equal(m, O.mrca_bulk(pairs)) and np.array_equal(tm, O.mrca_bulk(tri))
        dev.close()


def test_argument_errors_on_a_live_handle(gopher_flat):
    dev = _capi.DeviceTree(gopher_flat.parent, gopher_flat.distance)
    for name, value in (("pairs_per_lane", 1), ("tile_sort", 3), ("ladder_dynamic", 2), ("measure", 8), ("nope", 1)):
        with pytest.raises(ValueError):
            dev.set_option(name, value)
    with pytest.raises(ValueError):
        dev.triangle_host(np.arange(0, 10, 2), k_begin=0, k_count=11)        # 5 ids -> 10 pairs
    with pytest.raises(ValueError):
        dev.triangle_host(np.zeros((2, 2), dtype=np.int64))
    with pytest.raises(ValueError):
        dev.distances_host(np.array([[0, 2]]), True, False, out_dist=np.empty(3))
    with pytest.raises(ValueError):
        dev.distances_device(0, 5, 0, 0)                                      # NULL pairs, both outputs NULL
    info = dev.info()
    assert info["n_nodes"] == 29 and info["device_bytes"] > 0
    dev.close()
    with pytest.raises(_capi.HipBackendError):
        dev.info()                                                            # closed handle
    with pytest.raises(ValueError):
        _capi.DeviceTree(gopher_flat.parent, gopher_flat.distance[:-1])
    with pytest.raises(_capi.HipBackendError):
        _capi.DeviceTree(gopher_flat.parent, gopher_flat.distance, device=99)
