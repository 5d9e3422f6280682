"""Host-side logic of the multi-device handle and of the process model, no GPU needed:
the chunk -> device map of the host path (st_host_chunk_plan / st_host_chunk_owner, the same
arithmetic run_pipe uses) and the fork guard."""
import os

import numpy as np
import pytest

from suchtree_amd import _capi
from suchtree_amd.exceptions import HipBackendError


@pytest.mark.parametrize("n", [1, 2047, 262_144, 1_000_003, 4_194_304, 100_000_000, 4_999_950_000])
@pytest.mark.parametrize("n_dev", [1, 2, 3, 8])
def test_chunk_map_covers_the_batch_exactly_once(n, n_dev):
    chunks = _capi.host_chunk_map(n, n_dev)
    pos = 0
    for c, (dev, first, m) in enumerate(chunks):
        assert first == pos and m > 0          # contiguous, in order, no gaps, no overlap
        assert dev == c % n_dev                # dealt round-robin
        pos += m
    assert pos == n
    sizes = {m for _, _, m in chunks[:-1]}
    assert len(sizes) <= 1                     # one chunk size (the last may be short)
    assert max(m for _, _, m in chunks) <= 1 << 22
    if n_dev > 1 and n >= n_dev * (1 << 18):
        per_dev = np.bincount([d for d, _, _ in chunks], minlength=n_dev)
        assert per_dev.min() >= 1              # nobody idles on a batch worth dealing


def test_chunk_plan_argument_errors():
    with pytest.raises(ValueError):
        _capi.host_chunk_map(-1, 1)
    with pytest.raises(ValueError):
        _capi.host_chunk_map(10, 0)


def test_fork_after_gpu_use_is_refused_with_a_clear_error(monkeypatch):
    """A child forked after the parent initialised the GPU must get HipBackendError, not a
    hang (the reference's fork-pool recipe, docs/examples/SuchTree_examples.md:462-497)."""
    monkeypatch.setattr(_capi, "_gpu_pid", os.getpid())    # as if this process had uploaded a tree
    _capi._check_fork()                                    # same process: fine
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                           # child: must refuse before any HIP call
        os.close(r)
        try:
            _capi.DeviceTree(np.array([-1], np.int32), np.array([0], np.float32))
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
    assert out.startswith(b"refused:") and b"spawn" in out and b"forked" in out


def test_lazy_upload_keeps_the_fork_pool_recipe_usable():
    """Nothing touches the GPU until the first query, so trees created at module level can be
    inherited by forked workers that do their own upload."""
    from suchtree_amd import SuchTree
    before = _capi._gpu_pid
    T = SuchTree("((A:1,B:2):0.5,C:3);")
    assert T._dev_tree is None and _capi._gpu_pid == before


def test_recycle_pool_lends_recycles_and_respects_its_budget():
    """_capi.RecyclePool (no GPU): a block is on loan while the array or any view of it lives, comes back
    afterwards, is handed out again, and the budget holds."""
    import gc
    pool = _capi.RecyclePool(budget_bytes=192 << 20)
    a = pool.array(5_000_000, np.float64)               # 40 MB -> 64 MiB class
    assert a.shape == (5_000_000,) and a.dtype == np.float64 and pool.total == 64 << 20
    a[:] = 3.0
    view, ptr = a[10:20], a.ctypes.data
    del a
    gc.collect()
    assert not pool._free                              # the view keeps the block on loan
    del view
    gc.collect()
    assert list(pool._free) == [64 << 20] and len(pool._free[64 << 20]) == 1
    b = pool.array(4_500_000, np.float64)              # same size class: recycled, nothing new
    assert b.ctypes.data == ptr and pool.total == 64 << 20
    assert pool.array(10, np.float64) is None          # below 32 MiB: glibc recycles those itself
    c = pool.array(40_000_000, np.int32)               # 160 MB -> 192 MiB class: would exceed the budget
    assert c is None and pool.total == 64 << 20
    assert _capi.RecyclePool._size_class(3 << 20) == 4 << 20 and _capi.RecyclePool._size_class(100 << 20) == 128 << 20
    del b
    gc.collect()
    pool.trim()
    assert pool.total == 0 and not pool._free

def test_recycle_pool_of_ordinary_memory():
    """_capi.RecyclePool (the default home of result arrays of 32 MiB and more): on loan while the
    array or a view lives, handed out again afterwards (same memory), budget and trim."""
    import gc

    from suchtree_amd import _capi
    P = _capi.RecyclePool(budget_bytes=300 << 20)
    assert P.array(1000, np.float64) is None                       # small arrays: numpy's own business
    a = P.array(5_000_000, np.float64)
    assert a.shape == (5_000_000,) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    ptr = a.ctypes.data
    a[:] = 1.5
    view = a[10:20]
    del a
    gc.collect()
    assert not P._free                                             # the view keeps the block on loan
    assert view[0] == 1.5
    del view
    gc.collect()
    assert sum(len(v) for v in P._free.values()) == 1
    b = P.array(4_500_000, np.float64)                             # same size class: the same block
    assert b.ctypes.data == ptr
    c = P.array(9_000_000, np.int32)
    assert c is not None and c.ctypes.data != ptr and c.dtype == np.int32
    assert P.array(40_000_000, np.float64) is None                 # over budget: caller falls back to np.empty
    del b, c
    gc.collect()
    P.trim()
    assert P.total == 0 and not P._free
    assert _capi.RecyclePool(budget_bytes=0).array(50_000_000, np.float64) is None      # switched off


def test_table_budget_plan_drops_tables_in_the_stated_order(ml_arrays):
    """st_host_table_plan (no GPU): under a shrinking budget the accelerator tables go in the order of the ST_TABLE_*
    bits -- a smaller budget never keeps a table a larger one dropped --, the device bytes stay within the budget down
    to the floor (28 bytes per node), and below the records the walk family takes over."""
    from suchtree_amd import _capi, synth
    order = [name for _, name in _capi.DROPPED_TABLES]
    parent, dist, _ = ml_arrays
    trees = [("ml.tree", parent, dist), ("2^16 balanced", *synth.balanced_tree(16)), ("random 40k", *synth.random_binary_tree_levels(40_000, seed=2))]
    for what, par, dst in trees:
        full = _capi.host_table_plan(par, dst)
        assert full["dropped_tables"] == [] and full["family"] == "canopy", (what, full)
        floor = 28 * len(par) + 64
        seen_walk = False
        prev = set()
        mb = full["device_bytes"] / 2**20 * 1.01
        while mb > floor / 2**20 * 0.5:
            plan = _capi.host_table_plan(par, dst, table_mb=mb)
            dropped = plan["dropped_tables"]
            assert dropped == [t for t in order if t in dropped], (what, mb, dropped)      # listed in drop order
            assert prev <= set(dropped), (what, mb, prev, dropped)                           # monotone
            if mb * 2**20 >= floor:
                assert plan["device_bytes"] <= mb * 2**20, (what, mb, plan)
            else:
                assert plan["device_bytes"] <= floor + 4096 and plan["family"] == "walk", (what, mb, plan)
            if "canopy" in dropped:
                assert plan["family"] == "walk" and {"rec_i", "rec_a4", "ranks"} <= set(dropped)
                seen_walk = True
            else:
                assert plan["family"] == "canopy"
            prev = set(dropped)
            mb *= 0.8
        assert seen_walk and {"lineage_len", "lineage_sum", "tree_rmq", "rec_i", "canopy"} <= prev | {"lineage_len", "lineage_sum"}, (what, prev)
    # the walk family asked for: nothing of the canopy family is "dropped", the optional walk tables are
    plan = _capi.host_table_plan(parent, dist, strategy="walk", table_mb=4)
    assert plan["family"] == "walk" and "canopy" not in plan["dropped_tables"] and plan["device_bytes"] <= 4 * 2**20


def test_deepest_ladder_image_leaves_room_for_the_kernels_flags():
    """The scalar ladder kernel's LDS is the image (16 bytes per canopy node) plus 32 bytes of flags; a deep canopy is rebuilt
    with at most kDeepCanopyNodes nodes, and that many must fit the 160 KiB of a CU (round 4's advisor: 10240 nodes asked for
    163872 bytes and the launch failed).  The library asserts it at compile time; this reads the constants back."""
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "suchtree_amd", "csrc")
    nodes = int(re.search(r"constexpr int kDeepCanopyNodes = (\d+);", open(os.path.join(csrc, "host_path.h")).read()).group(1))
    geo = open(os.path.join(csrc, "launch_geometry.h")).read()
    flags = int(re.search(r"constexpr size_t kLadderFlagBytes = (\d+);", geo).group(1))
    assert "kLdsBytesPerCu = 160 * 1024" in geo and "static_assert(st::ladder_kernel_lds_bytes(kDeepCanopyNodes) <= st::kLdsBytesPerCu" in open(os.path.join(csrc, "host_path.h")).read()
    assert nodes * 16 + flags <= 160 * 1024 < (nodes + 2) * 16 + flags
    assert "ladder_kernel_lds_bytes(t->canopy_nodes) <= kLdsBytesPerCu" in open(os.path.join(csrc, "launch_policy.h")).read()
