"""The N>1 path on CPU: gloo process groups of 2, 3 and 4 ranks, pairs sharded across
ranks, result slices all-gathered.  The compute step is the oracle (a checker,
injected) because the product path refuses to run without a GPU; what is under
test is the partitioning and the gather, which are the same code on RCCL."""
import os
import socket

import numpy as np
import pytest

from suchtree_amd import sharding


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 12345):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_triangle_rows():
    m = 2000
    k = np.arange(m * (m - 1) // 2)
    i = sharding.triangle_row_of(k)
    j = k - i * (i - 1) // 2
    assert np.all((j >= 0) & (j < i)) and i.max() == m - 1 and i.min() == 1
    big = np.array([4_999_949_999, 4_999_850_001, 4_999_850_000], dtype=np.int64)   # 100k-leaf triangle
    assert sharding.triangle_row_of(big).tolist() == [99_999, 99_999, 99_998]
    lo, hi = sharding.triangle_shard_bounds(100_000, 8, 7)
    assert hi == 4_999_950_000 and hi - lo in (624_993_750, 624_993_749)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    import torch.distributed as dist
    from oracle.oracle import OracleTree
    from suchtree_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parent, dist_ = synth.balanced_tree(9)
        O = OracleTree(parent, dist_)
        pairs = np.random.default_rng(0).integers(0, len(parent), (n_pairs, 2))
        compute = lambda p: (O.distances(p), O.mrca_bulk(p))   # noqa: E731
        d, m = sharding.distances_sharded(None, pairs, compute=compute)
        ds, ms, (lo, hi) = sharding.distances_sharded(None, pairs, gather=False, compute=compute)
        ok = (np.array_equal(d.view(np.int64), O.distances(pairs).view(np.int64))
              and np.array_equal(m, O.mrca_bulk(pairs))
              and np.array_equal(ds, d[lo:hi]) and np.array_equal(ms, m[lo:hi])
              and (lo, hi) == sharding.shard_bounds(n_pairs, world, rank))
        q.put((rank, bool(ok), hi - lo))
    finally:
        dist.destroy_process_group()


def _worker_run_sharded(rank, world, port, n_pairs, chunks, root, q, root_share=None, packed=False):
    """bench.py's step, verbatim (sharding.run_sharded), on gloo/CPU tensors with the oracle
    injected as the compute step: slices, pieces, float32 wire format, assembly on the root."""
    import torch
    import torch.distributed as dist
    from oracle.oracle import OracleTree
    from suchtree_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parent, dist_ = synth.balanced_tree(9)
        O = OracleTree(parent, dist_)
        pairs = np.random.default_rng(5).integers(0, len(parent), (n_pairs, 2))
        if packed:
            pairs[::7, 0] = pairs[::7, 1]      # (x, x): every node id of the tree turns up as an MRCA, the largest included
        plan = sharding.ShardPlan(n_pairs, world, rank, chunks=chunks, root=root, root_share=root_share, align=4 if packed else 1)
        out_d, out_m, wire_d, wire_m = sharding.sharded_buffers(plan, packed_ids=packed)
        calls = []

        def compute(lo, hi, dst_d, dst_m):
            calls.append((lo, hi, dst_d.dtype))
            dst_d.copy_(torch.from_numpy(O.distances(pairs[lo:hi])).to(dst_d.dtype))
            ids = torch.from_numpy(O.mrca_bulk(pairs[lo:hi]))
            if dst_m.dtype == torch.uint8:      # the packed wire format: 24 bits per id, whole dwords (the kernels' stores)
                assert packed and rank != root and dst_m.numel() == sharding.packed_bytes(hi - lo) and dst_m.data_ptr() % 4 == 0
                dst_m.fill_(0)
                sharding.pack_mrca24(ids, dst_m)
            else:
                dst_m.copy_(ids)

        for _ in range(2):     # twice: buffers are reused from pass to pass, as in the bench loop
            sharding.run_sharded(plan, compute, out_d, out_m, wire_d, wire_m)
        ok = sum(hi - lo for lo, hi, _ in calls) == 2 * (plan.bounds(rank)[1] - plan.bounds(rank)[0])
        if rank == root:
            ok = ok and np.array_equal(out_d.numpy().view(np.int64), O.distances(pairs).view(np.int64))
            ok = ok and np.array_equal(out_m.numpy(), O.mrca_bulk(pairs))
            ok = ok and all(dt == torch.float64 for _, _, dt in calls)
        else:
            ok = ok and out_d.numel() == 0 and all(dt == torch.float32 for _, _, dt in calls)
        q.put((rank, bool(ok), len(calls)))
    finally:
        dist.destroy_process_group()


def _worker_run_allgather(rank, world, port, n_pairs, chunks, packed, q):
    """sharding.run_allgather on gloo/CPU tensors, the oracle as the compute step: every rank ends with the whole result."""
    import torch
    import torch.distributed as dist
    from oracle.oracle import OracleTree
    from suchtree_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parent, dist_ = synth.balanced_tree(9)
        O = OracleTree(parent, dist_)
        pairs = np.random.default_rng(6).integers(0, len(parent), (n_pairs, 2))
        pairs[::5, 0] = pairs[::5, 1]
        plan = sharding.ShardPlan(n_pairs, world, rank, chunks=chunks, align=4)
        out_d, out_m, wire_d, wire_m = sharding.sharded_buffers(plan, packed_ids=packed, all_ranks=True)
        calls = []

        def compute(lo, hi, dst_d, dst_m):
            calls.append((lo, hi))
            assert dst_d.dtype == torch.float32
            dst_d.copy_(torch.from_numpy(O.distances(pairs[lo:hi])).to(torch.float32))
            ids = torch.from_numpy(O.mrca_bulk(pairs[lo:hi]))
            if packed:
                assert dst_m.dtype == torch.uint8 and dst_m.numel() == sharding.packed_bytes(hi - lo) and dst_m.data_ptr() % 4 == 0
                dst_m.fill_(0)
                sharding.pack_mrca24(ids, dst_m)
            else:
                dst_m.copy_(ids)

        for _ in range(2):
            sharding.run_allgather(plan, compute, out_d, out_m, wire_d, wire_m)
        lo, hi = plan.bounds(rank)
        ok = sum(b - a for a, b in calls) == 2 * (hi - lo)
        ok = ok and np.array_equal(out_d.numpy().view(np.int64), O.distances(pairs).view(np.int64))
        ok = ok and np.array_equal(out_m.numpy(), O.mrca_bulk(pairs))
        q.put((rank, bool(ok), len(calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs,chunks,packed", [(2, 1001, 3, True), (2, 4, 4, True), (3, 1001, 4, False), (3, 1000, 1, True),
                                                         (4, 1003, 4, True), (4, 3, 2, False), (4, 1001, 3, False)])
def test_run_allgather_on_gloo(world, n_pairs, chunks, packed):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_run_allgather, args=(r, world, port, n_pairs, chunks, packed, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [r[1] for r in res] == [True] * world, res


_CASES = [
    (2, 1001, 3, 0, None), (2, 4, 4, 0, None), (2, 777, 1, 1, None), (2, 1001, 4, 0, 0.8), (2, 1001, 2, 1, 0.37), (2, 50, 3, 0, 1.0),
    # more than one peer: the root posts receives from several ranks per piece, peers are indexed around the root
    (3, 1001, 4, 0, None), (3, 1001, 1, 1, 0.5), (3, 1000, 4, 2, 0.31), (4, 1001, 4, 0, 0.5), (4, 1003, 1, 2, None),
    (4, 1003, 4, 1, 0.41), (4, 3, 4, 0, None), (4, 1001, 3, 3, 0.9)]


# every case in the packed wire format (the default of bench.py), every other one in the int32 format as well
@pytest.mark.parametrize("world,n_pairs,chunks,root,root_share,packed",
                         [c + (True,) for c in _CASES] + [c + (False,) for c in _CASES[::2]])
def test_run_sharded_gloo(world, n_pairs, chunks, root, root_share, packed):
    """packed: MRCA ids travel as 24 bits each (7 bytes per pair with the float32 distance), pieces start on
    4-pair boundaries of their slice, the root unpacks piece by piece."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_run_sharded, args=(r, world, port, n_pairs, chunks, root, q, root_share, packed))
             for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert [r[1] for r in res] == [True] * world
    assert all(p.exitcode == 0 for p in procs)


def _worker_measure(rank, world, port, root, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank passes a different kernel rate: the root's must win, and everyone must agree
        share, link, k = sharding.measure_root_share(world, rank, 1e9 * (rank + 1), nbytes=1 << 20, root=root)
        q.put((rank, share, link, k))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,root", [(2, 0), (3, 0), (4, 0), (4, 2)])
def test_measure_root_share_gloo(world, root):
    """bench.py's calibration step (batched point-to-point transfers from every peer into the root at
    once + broadcast of the root's decision)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_measure, args=(r, world, port, root, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    s0, l0, k0 = res[0][1:]
    assert all(r[1:] == (s0, l0, k0) for r in res)
    assert k0 == 1e9 * (root + 1) and l0 > 0 and 1.0 / world <= s0 <= 0.95
    assert s0 == min(0.95, max(1.0 / world, sharding.balanced_root_share(world, k0, l0)))


def test_balanced_root_share():
    # wire 8 B/pair at 60 GB/s = 133 ps, kernel 33.7 ps/pair: the root keeps 80 % at 2 GPUs, 36 % at 8
    s2 = sharding.balanced_root_share(2, 1 / 33.7e-12, 60e9)
    s8 = sharding.balanced_root_share(8, 1 / 33.7e-12, 60e9)
    assert abs(s2 - 0.798) < 0.005 and abs(s8 - 0.361) < 0.005 and sharding.balanced_root_share(1, 1e9, 1e9) == 1.0
    plan = sharding.ShardPlan(100_000_000, 8, 3, root=0, root_share=s8)
    sizes = [plan.bounds(g)[1] - plan.bounds(g)[0] for g in range(8)]
    assert sum(sizes) == 100_000_000 and sizes[0] == round(s8 * 1e8) and max(sizes[1:]) - min(sizes[1:]) <= 1
    # root and peers finish together under the model
    t_root = sizes[0] * 33.7e-12
    t_peer = sizes[1] * 8 / 60e9
    assert abs(t_root - t_peer) / t_root < 0.01


def test_pack_and_unpack_of_24_bit_ids():
    import torch
    ids = torch.tensor([0, 1, 255, 256, 65535, 65536, 0xFFFFFE, -1, 12345678, -1, 7], dtype=torch.int32)
    buf = torch.full((sharding.packed_bytes(len(ids)) + 5,), 0xAB, dtype=torch.uint8)
    sharding.pack_mrca24(ids, buf)
    assert bytes(buf[:6].tolist()) == bytes([0, 0, 0, 1, 0, 0]) and bytes(buf[21:24].tolist()) == b"\xff\xff\xff"
    assert all(int(b) == 0xAB for b in buf[3 * len(ids):])
    back = torch.empty(len(ids), dtype=torch.int32)
    sharding.unpack_mrca24(buf, back)
    assert torch.equal(back, ids)
    assert [sharding.packed_bytes(k) for k in (0, 1, 2, 3, 4, 5)] == [0, 4, 8, 12, 12, 16]


def test_shard_plan_pieces_tile_the_batch():
    cases = ((1, None, 0), (3, None, 0), (4, None, 0), (4, 0.45, 0), (3, 0.9, -1), (2, 1.0, 0))
    for n in (0, 5, 1000, 12345):
        for world in (1, 2, 8):
            for chunks, share, root in cases:
                for align in (1, 4):      # 4: pieces start on 4-pair boundaries of their slice (packed 24-bit ids)
                    seen = []
                    for g in range(world):
                        plan = sharding.ShardPlan(n, world, g, chunks=chunks, root=root % world, root_share=share, align=align)
                        assert plan.pieces(g)[0][0] == plan.bounds(g)[0] and plan.pieces(g)[-1][1] == plan.bounds(g)[1]
                        assert all((lo - plan.bounds(g)[0]) % align == 0 for lo, _ in plan.pieces(g))
                        seen += plan.pieces(g)
                    assert seen[0][0] == 0 and seen[-1][1] == n
                    assert all(a[1] == b[0] for a, b in zip(seen, seen[1:]))


@pytest.mark.parametrize("n_pairs", [1001, 4])
def test_world_size_2_gloo(n_pairs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert [r[1] for r in res] == [True, True]
    assert sum(r[2] for r in res) == n_pairs
    assert all(p.exitcode == 0 for p in procs)
