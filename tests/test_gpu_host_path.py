"""Host-buffer path on the GPU: multi-device handle (devices=[0] is all a 1-GPU box can run),
fresh vs reused result arrays, the separate fault words of the host and device-pointer paths,
and the fork guard after a real upload."""
import ctypes
import os

import numpy as np
import pytest

from conftest import assert_bits_equal, golden_path
from oracle.oracle import OracleTree
from suchtree_amd import SuchTree, _capi, synth
from suchtree_amd.exceptions import HipBackendError, InvalidNodeError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ml_arrays():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ml_tree.npz"))
    return z["parent"], z["distance"]


@pytest.fixture(scope="module")
def tree17():
    parent, dist = synth.balanced_tree(17)
    return parent, dist, OracleTree(parent, dist)


def test_multi_device_handle_on_one_gpu_matches_the_oracle(tree17):
    parent, dist, O = tree17
    T = SuchTree((parent, dist), devices=[0])
    info = T.device_info()
    assert info["n_devices"] == 1 and T._device_tree().devices == [0]
    rng = np.random.default_rng(11)
    for n in (1000, 300_000, 5_000_001):       # mailbox, one chunk, several chunks
        pairs = rng.integers(0, len(parent), (n, 2))
        d, m = T.distances_and_ancestors_bulk(pairs)
        k = min(n, 400_000)
        assert_bits_equal(d[:k], O.distances(pairs[:k]))
        assert np.array_equal(m[:k], O.mrca_bulk(pairs[:k]))
        assert_bits_equal(d[-k:], O.distances(pairs[-k:]))
    ids = np.asarray(T.leaf_node_ids[:700], dtype=np.int64)
    tri, _ = T._device_tree().triangle_host(ids)
    # same enumeration as linked_distances: k = i(i-1)/2 + j, pair (ids[j], ids[i])
    i, j = np.tril_indices(len(ids), -1)
    assert_bits_equal(tri, O.distances(np.stack([ids[j], ids[i]], 1)))
    with pytest.raises(InvalidNodeError) as e:
        bad = rng.integers(0, len(parent), (600_000, 2))
        bad[123_456, 1] = len(parent) + 7
        T.distances_bulk(bad)
    assert e.value.node_id == len(parent) + 7
    with pytest.raises(HipBackendError):
        SuchTree((parent, dist), devices=[0, 99]).to_device()
    with pytest.raises(ValueError):
        SuchTree((parent, dist), devices=[0, 0]).to_device()


def test_fresh_and_reused_result_arrays_agree(tree17):
    parent, dist, O = tree17
    dev = SuchTree((parent, dist)).to_device()._device_tree()
    pairs = np.random.default_rng(12).integers(0, len(parent), (9_000_000, 2))
    d0, m0 = dev.distances_host(pairs, True, True)                      # fresh arrays (pre-faulted by the pool)
    d1, m1 = np.full(len(pairs), -1.0), np.full(len(pairs), -7, np.int32)
    dev.distances_host(pairs, True, True, out_dist=d1, out_mrca=m1)     # resident arrays
    assert_bits_equal(d0, d1)
    assert np.array_equal(m0, m1)
    k = 300_000
    assert_bits_equal(d0[:k], O.distances(pairs[:k]))
    # odd offsets: result views that start in the middle of a page / a 16-byte line
    big_d, big_m = np.zeros(len(pairs) + 3), np.zeros(len(pairs) + 3, np.int32)
    dev.distances_host(pairs, True, True, out_dist=big_d[3:], out_mrca=big_m[1:-2])
    assert_bits_equal(big_d[3:], d0)
    assert np.array_equal(big_m[1:-2], m0) and big_d[:3].tolist() == [0, 0, 0] and big_m[0] == 0
    # int32 C-order, strided and wide-id inputs take the other pack branches
    d2, _ = dev.distances_host(pairs.astype(np.int32), True, False)
    assert_bits_equal(d2, d0)
    d3, _ = dev.distances_host(pairs[::2][:, ::-1], True, False)
    assert_bits_equal(d3[:k], O.distances(np.ascontiguousarray(pairs[::2][:, ::-1])[:k]))
    wide = pairs.copy()
    wide[4_321_000, 0] = 2**40 + 5
    wide[17, 1] = -(2**35)
    with pytest.raises(InvalidNodeError) as e:
        dev.distances_host(wide, True, False)
    assert e.value.node_id == 2**40 + 5          # the exact id, not the int32 clamp


def test_24_bit_ids_on_the_wire_and_int32_ids_agree(tree17):
    """Trees with fewer than 2^24 nodes ship host-path ids as 24 bits each (6 bytes per pair over the link; the
    range check is the packing step's).  Same results and the same reported id as the int32 wire format (option
    wire48 = 0), for C-order int64, int32 and strided views, batches that are no multiple of the packing granule,
    and every kind of bad id: >= n_nodes, >= 2^24, beyond int32, negative, several at once."""
    parent, dist, O = tree17
    n_nodes = len(parent)
    dev = _capi.DeviceTree(parent, dist)
    rng = np.random.default_rng(48)
    for n in (1_000_003, 262_145, 40_001):
        pairs = rng.integers(0, n_nodes, (n, 2))
        want_d, want_m = O.distances(pairs[:50_000]), O.mrca_bulk(pairs[:50_000])
        views = {"int64": pairs, "int32": pairs.astype(np.int32), "columns swapped": pairs[:, ::-1],
                 "every other row": np.concatenate([pairs, pairs], 1).reshape(-1, 2)[::2]}
        got = {}
        for w48 in (1, 0):
            dev.set_option("wire48", w48)
            for name, v in views.items():
                d, m = dev.distances_host(v, True, True)
                got[(w48, name)] = (d, m)
                if name in ("int64", "int32", "every other row"):
                    assert_bits_equal(d[:50_000], want_d, "%s wire48=%d n=%d" % (name, w48, n))
                    assert np.array_equal(m[:50_000], want_m)
        for name in views:
            assert_bits_equal(got[(1, name)][0], got[(0, name)][0], name)
            assert np.array_equal(got[(1, name)][1], got[(0, name)][1]), name
        # bad ids: the id the reference would report (the largest one >= n_nodes, else the smallest negative one)
        cases = [({5: n_nodes}, n_nodes), ({5: n_nodes + 7, 900: 2**24 + 3}, 2**24 + 3), ({77: 2**24 - 1}, 2**24 - 1),
                 ({n - 1: 2**40 + 5, 17: -(2**35)}, 2**40 + 5), ({3: -1}, -1), ({3: -1, n - 2: -9}, -9),
                 ({1000: -4, 2000: n_nodes + 1}, n_nodes + 1)]
        for bad_ids, reported in cases:
            bad = pairs.copy()
            for k, v in bad_ids.items():
                bad[k, k & 1] = v
            for w48 in (1, 0):
                dev.set_option("wire48", w48)
                with pytest.raises(InvalidNodeError) as e:
                    dev.distances_host(bad, True, True)
                assert e.value.node_id == reported, (bad_ids, w48, e.value.node_id)
    dev.set_option("wire48", 1)
    dev.close()


def test_24_bit_mrca_ids_on_the_way_back(tree17, ml_arrays):
    """Trees with fewer than 2^24 nodes: MRCA ids come back over the link as 24 bits each (7 bytes per pair with the
    float32 distance; option wire24 = 0: int32).  Every kernel the host path launches assembles the packed stream
    itself (k_canopy_ilp, k_canopy, k_walk, k_mrca_ranks, both tile-sorted kernels' key phases; the staged form of
    the tile-sorted canopy kernel packs in its copy kernel; pairs under one portal write their three bytes singly):
    same bits as the int32 form and as the oracle, for batch sizes that are no multiple of a quad, MRCA ids alone,
    generated pairs."""
    rng = np.random.default_rng(2424)
    ml_parent, ml_dist = ml_arrays
    sizes = (8_193, 40_001, 262_147, 1_000_003)

    def check(dev, pairs, want_d, want_m, what):
        for n in sizes:
            got = {}
            for w24 in (1, 0):
                dev.set_option("wire24", w24)
                d, m = dev.distances_host(pairs[:n], True, True)
                m_only = dev.distances_host(pairs[:n], False, True)[1]
                got[w24] = (d, m, m_only)
                k = min(n, len(want_m))
                assert_bits_equal(d[:k], want_d[:k], "%s wire24=%d n=%d" % (what, w24, n))
                assert np.array_equal(m[:k], want_m[:k]), (what, w24, n, np.flatnonzero(m[:k] != want_m[:k])[:8])
                assert np.array_equal(m_only[:k], want_m[:k]), (what, w24, n, "MRCA ids alone")
            assert_bits_equal(got[1][0], got[0][0], what)
            assert np.array_equal(got[1][1], got[0][1]) and np.array_equal(got[1][2], got[0][2]), (what, n)
        dev.set_option("wire24", 1)

    # shallow canopy: the predicated kernel in its forms, the scalar kernel, the walk family
    parent, dist, O = tree17
    n_nodes = len(parent)
    pairs = rng.integers(0, n_nodes, (sizes[-1], 2))
    pairs[5] = (n_nodes - 1, n_nodes - 1)                  # the largest id of the tree as an MRCA
    pairs[6] = (0, 0)
    k = 120_000
    want_d, want_m = O.distances(pairs[:k]), O.mrca_bulk(pairs[:k])
    dev = _capi.DeviceTree(parent, dist)
    check(dev, pairs, want_d, want_m, "2^17 leaves")
    dev.set_option("rec_a4", 0)
    check(dev, pairs, want_d, want_m, "2^17 leaves, 8-byte a side")
    dev.set_option("rec_a4", 1)
    dev.set_strategy("walk")
    check(dev, pairs, want_d, want_m, "2^17 leaves, walk family")
    dev.set_strategy("auto")
    ids = np.arange(0, 2 * 1500, 2, dtype=np.int64)
    i, j = np.tril_indices(len(ids), -1)
    tri_m = O.mrca_bulk(np.stack([ids[j], ids[i]], 1))
    for w24 in (1, 0):
        dev.set_option("wire24", w24)
        assert np.array_equal(dev.triangle_host(ids, want_dist=True, want_mrca=True)[1], tri_m)
    dev.set_option("wire24", 1)
    dev.close()

    # deep canopy: the tile-sorted kernels on the pinned slots, shared-portal pairs, the staged form
    n_nodes = len(ml_parent)
    a = rng.integers(0, n_nodes - 12, sizes[-1] // 2)
    near = np.stack([a, a + rng.integers(0, 12, a.size)], 1)           # mostly one portal
    far = rng.integers(0, n_nodes, (sizes[-1] - len(near), 2))
    pairs = np.concatenate([near, far])[rng.permutation(sizes[-1])].astype(np.int64)
    pairs[9] = (n_nodes - 1, n_nodes - 1)
    k = 60_000
    from conftest import oracle_both
    want_d, want_m = oracle_both(ml_parent, ml_dist, pairs[:k])
    dev = _capi.DeviceTree(ml_parent, ml_dist)
    base = {"tile_sort": 1, "ladder_scalar": 0, "prefer_walk_sorted": 0, "lineage_sums": 1}
    for opts in ({}, {"lineage_sums": 0}, {"tile_sort": 0}, {"tile_sort": 0, "ladder_scalar": 1}, {"prefer_walk_sorted": 1}):
        for name, v in dict(base, **opts).items():      # (every kernel by name: what the handle chose itself does not matter here)
            dev.set_option(name, v)
        check(dev, pairs, want_d, want_m, "ml.tree %s" % (opts or "tile-sorted canopy kernel"))
    dev.set_strategy("walk")
    check(dev, pairs, want_d, want_m, "ml.tree, walk family")
    dev.close()


def test_wire_formats_at_the_2_24_node_boundary():
    """A tree of 2^24 - 1 nodes (the largest the 24-bit formats serve; its largest id, 2^24 - 2, as an id and as an
    MRCA) and one of 2^24 + 1 nodes (int32 both ways: ids 2^24 - 1 and 2^24 come back whole)."""
    rng = np.random.default_rng(2**24)
    for n_leaves, packed in ((1 << 23, True), ((1 << 23) + 1, False)):
        parent, dist = (synth.balanced_tree(23) if packed else synth.random_binary_tree(n_leaves, seed=9))
        n_nodes = len(parent)
        assert n_nodes == 2 * n_leaves - 1 and (n_nodes <= 0xFFFFFF) == packed
        dev = _capi.DeviceTree(parent, dist)
        info = dev.info()
        assert (info["host_wire_bytes_in"], info["host_wire_bytes_out"]) == ((6, 7) if packed else (8, 8)), info
        pairs = rng.integers(0, n_nodes, (300_001, 2))
        top = np.arange(n_nodes - 40, n_nodes, dtype=np.int64)
        pairs[:40] = np.stack([top, top], 1)                  # MRCA = the id itself, up to n_nodes - 1
        pairs[40:80] = np.stack([top, top[::-1]], 1)
        O = OracleTree(parent, dist)
        k = 20_000
        want_d, want_m = O.distances(pairs[:k]), O.mrca_bulk(pairs[:k])
        assert want_m[39] == n_nodes - 1
        for w in (1, 0):
            dev.set_option("wire24", w)
            dev.set_option("wire48", w)
            d, m = dev.distances_host(pairs, True, True)
            assert_bits_equal(d[:k], want_d, "n_nodes=%d wire=%d" % (n_nodes, w))
            assert np.array_equal(m[:k], want_m)
            assert np.array_equal(dev.distances_host(pairs, False, True)[1][:k], want_m)
            bad = pairs.copy()
            bad[77, 1] = n_nodes
            with pytest.raises(InvalidNodeError) as e:
                dev.distances_host(bad, True, True)
            assert e.value.node_id == n_nodes
        dev.close()


def test_host_and_device_paths_do_not_share_a_fault_word(tree17):
    import torch
    parent, dist, O = tree17
    from suchtree_amd import torch_interop
    T = SuchTree((parent, dist)).to_device()
    n = len(parent)
    bad = torch.tensor([[0, 2], [n + 5, 4]] * 4000, dtype=torch.int64, device="cuda:0")
    d, m = torch_interop.distances_device(T, bad, check=False)           # leaves the device word set
    good = np.random.default_rng(13).integers(0, n, (100_000, 2))
    assert_bits_equal(T.distances_bulk(good), O.distances(good))         # host path: no spurious error
    with pytest.raises(InvalidNodeError) as e:                           # the device word is still there
        T._device_tree().fault_check(torch.cuda.current_stream().cuda_stream)
    assert e.value.node_id == n + 5
    T._device_tree().fault_check(torch.cuda.current_stream().cuda_stream)   # and is cleared by the report
    # out buffers of the torch entry points are validated before their pointers are taken
    ok = torch.tensor(good[:1000], device="cuda:0")
    with pytest.raises(ValueError):
        torch_interop.distances_device(T, ok, out_dist=torch.empty(999, dtype=torch.float64, device="cuda:0"))
    with pytest.raises(ValueError):
        torch_interop.distances_device(T, ok, out_mrca=torch.empty(1000, dtype=torch.int64, device="cuda:0"))
    with pytest.raises(ValueError):
        torch_interop.distances_device(T, ok, out_dist=torch.empty(1000, dtype=torch.float64))


def test_fork_after_upload_raises_in_the_child(tree17):
    parent, dist, _ = tree17
    T = SuchTree((parent, dist)).to_device()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        try:
            T.distances_bulk(np.array([[0, 2]]))
            msg = b"no error"
        except HipBackendError as e:
            msg = b"refused:" + str(e).encode()
        except BaseException as e:     # noqa: BLE001
            msg = b"other:" + repr(e).encode()
        os.write(w, msg)
        os._exit(0)
    os.close(w)
    out = os.read(r, 4096)
    os.waitpid(pid, 0)
    assert out.startswith(b"refused:") and b"spawn" in out
    assert T.distance(0, 2) == T.distance(0, 2)        # the parent's tree is untouched


def test_two_trees_share_one_staging_pipe(tree17):
    """One pinned staging pipe per device, whatever the number of trees (SuchLinkedTrees holds two)."""
    parent, dist, O = tree17
    p2, d2 = synth.balanced_tree(12)
    A, B = SuchTree((parent, dist)).to_device(), SuchTree((p2, d2)).to_device()
    rng = np.random.default_rng(14)
    pa, pb = rng.integers(0, len(parent), (700_000, 2)), rng.integers(0, len(p2), (700_000, 2))
    da, db = A.distances_bulk(pa), B.distances_bulk(pb)
    assert_bits_equal(da[:100_000], O.distances(pa[:100_000]))
    assert_bits_equal(db[:100_000], OracleTree(p2, d2).distances(pb[:100_000]))
    A.close()
    assert_bits_equal(B.distances_bulk(pb), db)        # the pipe outlives the first tree


def test_result_arrays_in_pinned_memory_are_ordinary_result_arrays(tree17):
    """Round 4 let the kernels write float64 + int32 straight into result arrays that were themselves pinned (half the
    rate of the staged form: removed).  A caller's pinned array is now filled like any other host array."""
    import torch
    parent, dist, O = tree17
    rng = np.random.default_rng(15)
    dev = _capi.DeviceTree(parent, dist)
    pairs = rng.integers(0, len(parent), (1_000_001, 2))
    out_d = torch.empty(len(pairs), dtype=torch.float64).pin_memory()
    out_m = torch.empty(len(pairs), dtype=torch.int32).pin_memory()
    d, m = dev.distances_host(pairs, True, True, out_dist=out_d.numpy(), out_mrca=out_m.numpy())
    assert_bits_equal(d, O.distances(pairs))
    assert np.array_equal(m, O.mrca_bulk(pairs))
    dev.close()


def test_caller_supplied_pinned_outputs_take_the_direct_path(tree17):
    import torch
    parent, dist, O = tree17
    dev = SuchTree((parent, dist)).to_device()._device_tree()
    pairs = np.random.default_rng(16).integers(0, len(parent), (2_500_000, 2))
    out_d = torch.empty(len(pairs) + 3, dtype=torch.float64).pin_memory()
    out_m = torch.empty(len(pairs), dtype=torch.int32)                 # pageable: mixed modes in one call
    pin_m = torch.empty(len(pairs) + 1, dtype=torch.int32).pin_memory()   # pinned, 4-byte-aligned view below
    out_d.fill_(-1.0)
    dev.distances_host(pairs, True, True, out_dist=out_d.numpy()[3:], out_mrca=out_m.numpy())
    assert out_d[:3].tolist() == [-1.0, -1.0, -1.0]
    assert_bits_equal(out_d.numpy()[3:][:400_000], O.distances(pairs[:400_000]))
    assert_bits_equal(out_d.numpy()[3:][-400_000:], O.distances(pairs[-400_000:]))
    assert np.array_equal(out_m.numpy()[-400_000:], O.mrca_bulk(pairs[-400_000:]))
    z = np.load(golden_path("ml_tree.npz"))                              # tile-sorted tree: staged copies into pinned outputs
    deep = SuchTree((z["parent"], z["distance"])).to_device()._device_tree()
    dp = np.random.default_rng(17).integers(0, len(z["parent"]), (len(pairs), 2))
    deep.distances_host(dp, True, True, out_dist=out_d.numpy()[3:], out_mrca=pin_m.numpy()[1:])
    OD = OracleTree(z["parent"], z["distance"])
    assert_bits_equal(out_d.numpy()[3:][-300_000:], OD.distances(dp[-300_000:]))
    assert np.array_equal(pin_m.numpy()[1:][-300_000:], OD.mrca_bulk(dp[-300_000:]))
    assert np.array_equal(pin_m.numpy()[1:][:300_000], OD.mrca_bulk(dp[:300_000]))
    bad = pairs.copy()
    bad[77, 0] = -5
    with pytest.raises(InvalidNodeError) as e:
        dev.distances_host(bad, True, False, out_dist=out_d.numpy()[3:])
    assert e.value.node_id == -5


def test_create_destroy_cycles_do_not_leak_device_or_pinned_memory(tree17):
    """Trees come and go (the per-device staging pipe is reference counted): HBM returns to its
    level and the pipe is rebuilt on demand."""
    import torch
    parent, dist, O = tree17
    pairs = np.random.default_rng(18).integers(0, len(parent), (400_000, 2))
    want = O.distances(pairs[:50_000])

    def cycle():
        T = SuchTree((parent, dist))
        d = T.distances_bulk(pairs)
        assert_bits_equal(d[:50_000], want)
        T.close()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "device memory shrank by %d MiB over 25 create/close cycles" % ((free0 - free1) >> 20)


def test_chunks_dealt_over_several_replicas(tree17, monkeypatch):
    """The multi-device code path on a one-GPU box: three replicas of the tree on device 0 (a
    testing aid, SUCHTREE_AMD_ALLOW_DUPLICATE_DEVICES), one host thread each, chunks dealt
    round-robin, fault words merged -- the results must not depend on who computed what."""
    monkeypatch.setenv("SUCHTREE_AMD_ALLOW_DUPLICATE_DEVICES", "1")
    parent, dist, O = tree17
    T = SuchTree((parent, dist), devices=[0, 0, 0])
    assert T.device_info()["n_devices"] == 3
    rng = np.random.default_rng(19)
    n = 3_300_001                                   # 13 chunks of 2^18 pairs over three replicas
    assert len(_capi.host_chunk_map(n, 3)) >= 9
    pairs = rng.integers(0, len(parent), (n, 2))
    d, m = T.distances_and_ancestors_bulk(pairs)
    for lo in (0, 1_000_000, n - 300_000):
        assert_bits_equal(d[lo:lo + 300_000], O.distances(pairs[lo:lo + 300_000]))
        assert np.array_equal(m[lo:lo + 300_000], O.mrca_bulk(pairs[lo:lo + 300_000]))
    ids = np.asarray(T.leaf_node_ids[:2600], dtype=np.int64)
    tri, _ = T._device_tree().triangle_host(ids)                     # 3.4e6 generated pairs, dealt too
    i, j = np.tril_indices(len(ids), -1)
    sel = rng.integers(0, len(i), 300_000)
    assert_bits_equal(tri[sel], O.distances(np.stack([ids[j[sel]], ids[i[sel]]], 1)))
    bad = pairs.copy()
    bad[2_900_000, 0] = len(parent) + 11            # lands in one replica's chunk ...
    bad[100, 1] = -3                                # ... and this one in another's
    with pytest.raises(InvalidNodeError) as e:
        T.distances_bulk(bad)
    assert e.value.node_id == len(parent) + 11      # max id too large wins over the negative one, as in the reference
    T.close()
