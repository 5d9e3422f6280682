"""bench.py's job -- root-share calibration, sharded step, barriers, max-over-ranks timing, deadline, the
JSON line with its N > 1 fields, CPU baseline and parity -- at world sizes 2, 3 and 4 on gloo/CPU tensors.
The backend injected here computes with the oracle (the product path refuses to run without a GPU); what is
under test is everything around the kernel, which is the same code under RCCL."""
import json
import os
import socket
import time

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class CpuBackend:
    """What bench.HipBackend is to the GPU, on CPU tensors with the oracle as the compute step."""
    name = "oracle-on-cpu (test)"
    calibration_bytes = 1 << 16

    def __init__(self, parent, dist, levels, rank, hang_rank=None):
        import torch
        from oracle.oracle import OracleTree
        self.torch = torch
        self.device = torch.device("cpu")
        self.O = OracleTree(parent, dist)
        self.n_leaves = 1 << levels
        self.n_nodes = len(parent)
        self.rank = rank
        self.hang_rank = hang_rank
        self._ms = 0.0
        self._resets = 0
        self.calls = []

    def info(self):
        return {"strategy": "canopy", "canopy_nodes": 0, "record_bytes": 64, "n_nodes": self.n_nodes}

    def make_pairs(self, n, seed):
        return self.torch.from_numpy(np.random.default_rng(seed).integers(0, self.n_leaves, (n, 2)) * 2)

    def bind(self, pairs):
        p = pairs.numpy()

        def compute(lo, hi, dst_d, dst_m):
            if self.hang_rank == self.rank:
                time.sleep(120)
            t0 = time.perf_counter()
            dst_d.copy_(self.torch.from_numpy(self.O.distances(p[lo:hi])).to(dst_d.dtype))
            ids = self.torch.from_numpy(self.O.mrca_bulk(p[lo:hi]))
            if dst_m.dtype == self.torch.uint8:      # the gather's packed wire format (what st_distances_device_wire writes)
                from suchtree_amd import sharding
                assert dst_m.numel() == sharding.packed_bytes(hi - lo)
                sharding.pack_mrca24(ids, dst_m)
            else:
                dst_m.copy_(ids)
            self._ms += (time.perf_counter() - t0) * 1e3
            self.calls.append((lo, hi))
        return compute

    def kernel_clock_reset(self):
        self._ms = 0.0
        if not self._resets:      # (the warmup's calls go; the gather sweep's later resets keep the timed region's)
            self.calls.clear()
        self._resets += 1

    def kernel_ms_total(self):
        return self._ms

    def synchronize(self):
        pass

    def fault_check(self):
        pass

    def kernel_rate(self, pairs, m):
        return 2.0e6 * (self.rank + 1)       # every rank a different figure: rank 0's must be the one used


def ShardPlanShare(sharding, n, world, chunks, share):
    lo, hi = sharding.ShardPlan(n, world, 0, chunks=chunks, root_share=share, align=4).bounds(0)
    return (hi - lo) / n


def _worker(rank, world, port, argv, q, hang_rank=None):
    import torch.distributed as dist
    import bench
    from suchtree_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = bench.parse(argv)
    parent, dist_ = synth.balanced_tree(args.levels)
    be = CpuBackend(parent, dist_, args.levels, rank, hang_rank)
    line = bench.run_job(args, be, dist, world, rank, parent, dist_)
    q.put((rank, json.dumps(line) if line is not None else None, list(be.calls)))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, argv, hang_rank=None, timeout=180):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q, hang_rank)) for r in range(world)]
    [p.start() for p in procs]
    return procs, q, timeout


@pytest.mark.parametrize("world,share,chunks,n,wire", [(2, "auto", 4, 30011, 7), (3, "auto", 4, 30011, 7), (4, "even", 1, 1003, 7),
                                                        (4, "0.5", 3, 30011, 8), (3, "0.9", 2, 7, 7), (2, "auto", 3, 30013, 8),
                                                        (8, "auto", 2, 30011, 7)])
def test_bench_job_line_on_gloo(world, share, chunks, n, wire):
    """wire = 7: float32 + 24-bit MRCA ids travel (packed by the compute step, unpacked on the root piece by piece);
    8: --wire-int32."""
    from suchtree_amd import sharding
    argv = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--pairs", str(n), "--levels", "9",
            "--chunks", str(chunks), "--root-share", share, "--cpu-seconds", "0.2", "--deadline", "120"] + (["--wire-int32"] if wire == 8 else [])
    procs, q, timeout = _run(world, argv)
    res = sorted(q.get(timeout=timeout) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [r[1] is not None for r in res] == [True] + [False] * (world - 1)      # one line, from rank 0
    d = json.loads(res[0][1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "parity_across_slices",
                "gather_ms", "gather_bytes_into_root", "root_share", "kernel_only_pairs_per_s", "checksum"):
        assert key in d, key
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["value"] > 0 and d["gather_ms"] >= 0 and d["config"]["pairs_per_step"] == n
    assert d["parity"]["distances_bit_exact"] and d["parity"]["mrca_bit_exact"]
    assert d["parity_across_slices"]["distances_bit_exact"] and d["parity_across_slices"]["mrca_bit_exact"]
    assert d["parity_across_slices"]["checked_pairs"] >= min(n, 1000)
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches_per_step"] == chunks
    # at N > 1 the block speaks for the slowest rank's kernels: its kernel time and its pairs
    slow = int(np.argmax(d["per_rank"]["kernel_ms"]))
    assert r["rank"] == slow and r["kernel_ms"] == d["per_rank"]["kernel_ms"][slow]
    assert r["pairs_per_launch"] == d["per_rank"]["pairs"][slow] // chunks
    assert d["cpu_baseline"]["host_cpus"] >= d["cpu_baseline"]["cores"]
    gm = d["gather_modes"]
    assert set(gm) == {"root", "allgather", "none"}
    for v in gm.values():      # one run separates kernel scaling from link limits: kernels alone and what each mode's transfers add
        assert v["kernel_ms_slowest_rank"] > 0 and v["gather_ms"] >= 0 and 0 < v["root_share"] <= 1
    assert abs(gm["root"]["root_share"] - d["root_share"]) < 1e-12
    # every rank computed exactly its slice, in `chunks` pieces per step, for the 2 timed steps
    if share == "auto":
        cal = d["root_share_calibration"]
        assert cal["kernel_pairs_per_s"] == 2.0e6 and cal["link_GBps_into_root_per_peer"] > 0
        assert abs(cal["projected_root_share_at_60GBps_per_link"] - sharding.balanced_root_share(world, 2.0e6, 60e9, wire)) < 1e-12
        assert abs(cal["calibrated_speedup"] * cal["root_share"] - 1.0) < 1e-9
        expect = min(0.95, max(1.0 / world, sharding.balanced_root_share(world, 2.0e6, cal["link_GBps_into_root_per_peer"] * 1e9, wire)))
        assert abs(expect - cal["root_share"]) < 1e-9      # (the line's own figure is used below: GB/s and back costs the last bit)
        expect = cal["root_share"]
    else:
        expect = None if share == "even" else float(share)
    covered = []
    for rank, _, calls in res:
        plan = sharding.ShardPlan(n, world, rank, chunks=chunks, root_share=expect, align=4)
        want = [pc for pc in plan.pieces(rank) if pc[1] > pc[0]] * 2
        assert [tuple(c) for c in calls][: len(want)] == want      # (then: a few steps of the two other gather modes, gather_modes)
        covered += want[: len(want) // 2]
        if rank == 0:
            assert abs(d["root_share"] - (plan.bounds(0)[1] - plan.bounds(0)[0]) / n) < 1e-12
            assert d["wire_bytes_per_pair"] == wire
            assert d["gather_bytes_into_root"] == wire * (n - (plan.bounds(0)[1] - plan.bounds(0)[0]))
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


@pytest.mark.parametrize("world,mode,chunks,n,wire", [(2, "allgather", 3, 30011, 7), (3, "allgather", 4, 30013, 8), (4, "allgather", 1, 1003, 7),
                                                     (2, "none", 2, 30011, 7), (3, "none", 1, 30013, 7), (4, "none", 4, 1003, 8),
                                                     (8, "allgather", 2, 30011, 7)])
def test_bench_gather_modes_on_gloo(world, mode, chunks, n, wire):
    """--gather allgather (the result assembled on every rank by grouped send/recv between all pairs of ranks) and
    --gather none (every rank keeps its slice; assembled once, untimed, for the parity check): even slices, the same line."""
    from suchtree_amd import sharding
    argv = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--pairs", str(n), "--levels", "9", "--gather", mode,
            "--chunks", str(chunks), "--cpu-seconds", "0.2", "--deadline", "120"] + (["--wire-int32"] if wire == 8 else [])
    procs, q, timeout = _run(world, argv)
    res = sorted(q.get(timeout=timeout) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    d = json.loads(res[0][1])
    assert d["gather"] == mode and d["config"]["gather"] == mode and d["n_gpus"] == world and d["scaling"] == "strong"
    # the other two modes are timed for a few steps after the timed region and ride in the same line
    gm = d["gather_modes"]
    assert set(gm) == {"root", "allgather", "none"} and gm[mode]["timed_region"] and abs(gm[mode]["pairs_per_s"] - d["value"]) < 1e-6 * d["value"]
    assert all(v["pairs_per_s"] > 0 and v["steps"] >= 1 for v in gm.values()) and sum(v["timed_region"] for v in gm.values()) == 1
    assert d["parity"]["distances_bit_exact"] and d["parity"]["mrca_bit_exact"]
    assert d["parity_across_slices"]["distances_bit_exact"] and d["parity_across_slices"]["mrca_bit_exact"]
    assert d["process_group"] == {"backend": "gloo", "world_size": world, "what": d["process_group"]["what"]}
    assert len(d["per_rank"]["kernel_ms"]) == world and sum(d["per_rank"]["pairs"]) == n
    assert abs(d["root_share"] - (sharding.shard_bounds(n, world, 0)[1]) / n) < 1e-12
    # the sweep's root mode runs with the calibrated root share (the balanced gather of the default run), and says which
    cal = d["root_share_calibration"]
    assert abs(gm["root"]["root_share"] - ShardPlanShare(sharding, n, world, chunks, cal["root_share"])) < 1e-12
    assert abs(gm[mode]["root_share"] - d["root_share"]) < 1e-12
    assert d["gather_bytes_into_root"] == (0 if mode == "none" else wire * (n - sharding.shard_bounds(n, world, 0)[1]))
    covered = []
    for rank, _, calls in res:
        plan = sharding.ShardPlan(n, world, rank, chunks=chunks, root_share=None, align=4)
        want = [pc for pc in plan.pieces(rank) if pc[1] > pc[0]] * 2
        assert [tuple(c) for c in calls][: len(want)] == want      # (then: the untimed assembly of --gather none, the other modes' steps)
        covered += want[: len(want) // 2]
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_a_hung_rank_ends_the_job_at_the_deadline():
    """Rank 1 never finishes its first piece: every rank must leave with code 3 when its deadline
    expires instead of waiting for the backend's own collective timeout."""
    argv = ["--gpus", "3", "--steps", "1", "--warmup", "0", "--pairs", "1000", "--levels", "9", "--root-share", "even",
            "--no-cpu-baseline", "--deadline", "4"]
    t0 = time.time()
    procs, q, _ = _run(3, argv, hang_rank=1)
    [p.join(timeout=90) for p in procs]
    assert time.time() - t0 < 80
    assert [p.exitcode for p in procs] == [3, 3, 3]


def test_weak_mode_line_on_gloo():
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "0", "--pairs", "501", "--levels", "9", "--weak", "--cpu-seconds", "0.1"]
    procs, q, timeout = _run(2, argv)
    res = sorted(q.get(timeout=timeout) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    d = json.loads(res[0][1])
    assert d["scaling"] == "weak" and d["config"]["pairs_per_step"] == 1002 and "gather_ms" not in d
    assert d["parity"]["distances_bit_exact"] and all(len(r[2]) == 2 for r in res)


def _worker_closing(rank, world, port, q):
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.time()
    if rank == 0:
        time.sleep(1.5)                                  # rank 0 is still busy with its CPU legs
        bench.closing_wait(dist, rank, 60, release=True)
    else:
        bench.closing_wait(dist, rank, 60)               # what run_job's peers do after the timed region
    waited = time.time() - t0
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, waited))


def test_peers_wait_on_the_store_until_rank_0_has_printed_its_line():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_closing, args=(r, 3, port, q)) for r in range(3)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[1] >= 1.2 and res[2] >= 1.2 and max(res.values()) < 30
