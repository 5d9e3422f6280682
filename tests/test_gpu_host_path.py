"""Host-buffer path on the GPU: multi-device handle (devices=[0] is all a 1-GPU box can run),
fresh vs reused result arrays, the separate fault words of the host and device-pointer paths,
and the fork guard after a real upload."""
import ctypes
import os

import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle.oracle import OracleTree
from suchtree_amd import SuchTree, _capi, synth
from suchtree_amd.exceptions import HipBackendError, InvalidNodeError

pytestmark = pytest.mark.gpu


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
