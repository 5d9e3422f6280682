"""bench.py's sharded step (sharding.run_sharded) with the product's kernels in several PROCESSES: every rank builds the
tree on the one GPU of the box, computes its pieces there -- peers in the gather's wire format (float32 + 24-bit ids,
packed by the kernels at the piece offsets the plan gives) -- and the pieces travel over gloo to the root, which
unpacks and widens them.  RCCL needs one GPU per rank, so the transport is gloo here; everything else (plan, pieces,
offsets, wire format, kernels, several processes with HIP) is what `bench.py --gpus N` runs."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, chunks, root, root_share, packed, q, allgather=False):
    import torch
    import torch.distributed as dist
    from suchtree_amd import _capi, sharding, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parent, dist_ = synth.skewed_tree(np.random.default_rng(3), 40_000, 0.6)
        tree = _capi.DeviceTree(parent, dist_, device=0)
        host = np.random.default_rng(5).integers(0, len(parent), (n_pairs, 2))
        host[::7, 0] = host[::7, 1]
        pairs = torch.from_numpy(host).cuda()
        plan = sharding.ShardPlan(n_pairs, world, rank, chunks=chunks, root=root, root_share=root_share, align=4 if packed else 1)
        out_d, out_m, wire_d, wire_m = sharding.sharded_buffers(plan, packed_ids=packed, all_ranks=allgather)      # CPU tensors (gloo)

        def compute(lo, hi, dst_d, dst_m):
            n = hi - lo
            f32 = dst_d.dtype == torch.float32
            dev_d = torch.empty(n, dtype=dst_d.dtype, device="cuda")
            if dst_m.dtype == torch.uint8:
                assert dst_m.numel() == sharding.packed_bytes(n)
                dev_m = torch.zeros(dst_m.numel(), dtype=torch.uint8, device="cuda")
                tree.distances_device_wire(pairs.data_ptr() + lo * 16, n, dev_d.data_ptr(), dev_m.data_ptr())
            else:
                dev_m = torch.empty(n, dtype=torch.int32, device="cuda")
                tree.distances_device(pairs.data_ptr() + lo * 16, n, dev_d.data_ptr(), dev_m.data_ptr(), f32=f32)
            torch.cuda.synchronize()
            dst_d.copy_(dev_d.cpu())
            dst_m.copy_(dev_m.cpu())

        for _ in range(2):
            if allgather:
                sharding.run_allgather(plan, compute, out_d, out_m, wire_d, wire_m)
            else:
                sharding.run_sharded(plan, compute, out_d, out_m, wire_d, wire_m)
        tree.fault_check()
        if rank == root or allgather:
            q.put((rank, out_d.numpy().copy(), out_m.numpy().copy()))
        else:
            q.put((rank, None, None))
        tree.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs,chunks,root,root_share,packed", [
    (2, 300_001, 4, 0, None, True), (3, 300_001, 3, 1, 0.5, True), (3, 100_003, 4, 0, 0.2, False), (4, 64, 4, 2, None, True)])
def test_sharded_step_in_processes_on_one_gpu(world, n_pairs, chunks, root, root_share, packed):
    import torch.multiprocessing as mp
    from conftest import oracle_both
    from suchtree_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, chunks, root, root_share, packed, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    parent, dist_ = synth.skewed_tree(np.random.default_rng(3), 40_000, 0.6)
    host = np.random.default_rng(5).integers(0, len(parent), (n_pairs, 2))
    host[::7, 0] = host[::7, 1]
    want_d, want_m = oracle_both(parent, dist_, host)
    got_d, got_m = res[root][1], res[root][2]
    assert np.array_equal(got_d.view(np.int64), want_d.view(np.int64)) and np.array_equal(got_m, want_m)
    assert all(r[1] is None for r in res if r[0] != root)


@pytest.mark.parametrize("world,n_pairs,chunks,packed", [(2, 300_001, 3, True), (3, 100_003, 4, False), (4, 70_001, 2, True)])
def test_allgather_step_in_processes_on_one_gpu(world, n_pairs, chunks, packed):
    """sharding.run_allgather (bench.py --gather allgather) with the product's kernels: every rank ends with the whole
    result; the packed ids of every slice sit in a region of their own of the wire buffer, written there by the kernels."""
    import torch.multiprocessing as mp
    from conftest import oracle_both
    from suchtree_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, chunks, 0, None, packed, q, True)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    parent, dist_ = synth.skewed_tree(np.random.default_rng(3), 40_000, 0.6)
    host = np.random.default_rng(5).integers(0, len(parent), (n_pairs, 2))
    host[::7, 0] = host[::7, 1]
    want_d, want_m = oracle_both(parent, dist_, host)
    for _, got_d, got_m in res:
        assert np.array_equal(got_d.view(np.int64), want_d.view(np.int64)) and np.array_equal(got_m, want_m)
