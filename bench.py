#!/usr/bin/env python3
"""Benchmark of the hot path: leaf-pair patristic distances (+ MRCA ids) per second
on BASELINE's 1M-leaf tree.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md section 8d "Config 3"): synthetic
perfectly balanced binary tree with 2^20 leaves (2,097,151 nodes, in-order ids),
uniform random leaf pairs, int64 (n,2) already resident in HBM when the timed
region starts; one step = one launch computing float64 distances AND int32 MRCA
ids for the rank's whole batch.  Weak scaling: every rank owns its own batch of
--pairs pairs (pairs are independent, no data-path collective); value = pairs of
all ranks / max-over-ranks time.  For N > 1 the driver launches this file with
torch.distributed.run, one rank per GPU; RCCL is only used for the barrier, the
max-reduction of the time and the (untimed, reported) gather of result shards.

One JSON line on stdout from rank 0; see README/DESIGN.md for the fields.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=100_000_000, help="pairs per rank per step")
    ap.add_argument("--levels", type=int, default=20, help="balanced tree with 2**levels leaves")
    ap.add_argument("--strategy", default="auto", choices=["auto", "canopy", "walk"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive end-to-end leg")
    return ap.parse_args()


def cpu_baseline(parent, dist, pairs_host, gpu_dist, gpu_mrca, seconds):
    """Oracle (CPU port of the reference algorithm) on a bounded sample of the same
    pairs, all host cores, fork-pool-style contiguous chunks.  Also the parity check
    of the timed GPU results on that sample."""
    from oracle.oracle import OracleTree
    O = OracleTree(parent, dist)
    cores = len(os.sched_getaffinity(0))
    probe = min(len(pairs_host), 200_000 * cores)
    t0 = time.perf_counter()
    O.distances_mt(pairs_host[:probe], cores)
    rate = probe / (time.perf_counter() - t0)
    n = int(min(len(pairs_host), max(probe, rate * seconds)))
    t0 = time.perf_counter()
    d = O.distances_mt(pairs_host[:n], cores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    O.distances(pairs_host[: min(n, 1_000_000)])
    rate_1 = min(n, 1_000_000) / (time.perf_counter() - t1)
    m_n = min(n, 2_000_000)
    m = O.mrca_bulk(pairs_host[:m_n])
    bit_exact = bool(np.array_equal(d.view(np.int64), gpu_dist[:n].view(np.int64)))
    mrca_exact = bool(np.array_equal(m, gpu_mrca[:m_n]))
    max_rel = float(np.max(np.abs(d - gpu_dist[:n]) / np.maximum(np.abs(d), 1e-300))) if n else 0.0
    return {
        "value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": "first %d pairs of rank 0's batch, oracle/suchtree_oracle.c (visited-list MRCA, 20-byte AoS), "
                  "%d pthreads on contiguous chunks" % (n, cores),
        "single_thread_value": rate_1,
    }, {"distances_bit_exact": bit_exact, "mrca_bit_exact": mrca_exact, "max_rel_err": max_rel,
        "checked_pairs": n}


def latest_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes, if a summary was committed."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))
    except Exception:
        return None


def main():
    args = parse()
    # Only the JSON line may reach stdout: RCCL prints a version banner to fd 1 when a
    # communicator is created, so everything else is sent to stderr until the line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist_

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL code path is used even with one rank,
    # so that it can be exercised on a 1-GPU box
    distributed = "RANK" in os.environ or world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from suchtree_amd import _capi, synth
    parent, dist = synth.balanced_tree(args.levels)
    n_leaves = 1 << args.levels
    tree = _capi.DeviceTree(parent, dist, device=local_rank, strategy=args.strategy)
    info = tree.info()

    # synthetic pair batch, generated on the device (leaf ids are the even ids)
    n = args.pairs
    gen = torch.Generator(device=device)
    gen.manual_seed(3 + rank)
    pairs = torch.randint(0, n_leaves, (n, 2), generator=gen, device=device, dtype=torch.int64) * 2
    out_d = torch.empty(n, dtype=torch.float64, device=device)
    out_m = torch.empty(n, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)

    def step():
        tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)

    def barrier():
        if distributed:
            dist_.barrier()

    for _ in range(args.warmup):
        step()
    tree.fault_check(stream.cuda_stream)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize(device)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step()
        ev[k][1].record(stream)
    torch.cuda.synchronize(device)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist_.all_reduce(t, op=dist_.ReduceOp.MAX)
        elapsed = float(t.item())
    tree.fault_check(stream.cuda_stream)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # algorithmic bytes per pair: 16 in + 8 + 4 out + 8 per edge of the path
    depth_t = torch.from_numpy(_depths(parent)).to(device)
    h = depth_t[pairs[:, 0]] + depth_t[pairs[:, 1]] - 2 * depth_t[out_m.long()]
    h_mean = float(h.double().mean().item())
    bytes_per_pair = 16 + 8 + 4 + 8 * h_mean
    checksum = float(out_d.sum().item())

    gather_ms = None
    if distributed and not args.no_gather:
        # the north star's "final gather" of result shards over xGMI (untimed, reported)
        all_d = torch.empty(world * n, dtype=torch.float64, device=device)
        all_m = torch.empty(world * n, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)
        barrier()
        g0 = time.perf_counter()
        dist_.all_gather_into_tensor(all_d, out_d)
        dist_.all_gather_into_tensor(all_m, out_m)
        torch.cuda.synchronize(device)
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert torch.equal(all_d[rank * n:(rank + 1) * n], out_d)
        del all_d, all_m

    if rank == 0:
        total_pairs = float(n) * world * args.steps
        value = total_pairs / elapsed
        achieved = bytes_per_pair * n / (kernel_ms * 1e-3) / 1e9
        traffic = latest_traffic()
        line = {
            "metric": "leaf-pair patristic distances/sec (+ MRCA ids/sec), 1M-leaf tree",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "balanced 2^%d-leaf tree (%d nodes), %d uniform random leaf pairs per GPU per step, "
                                   "int64 ids in HBM -> float64 distance + int32 MRCA id"
                                   % (args.levels, len(parent), n),
                       "pairs_per_gpu": n, "tree_levels": args.levels, "kernel_family": info["strategy"],
                       "canopy_nodes": info["canopy_nodes"], "record_bytes": info["record_bytes"],
                       "sharding": "pairs sharded across ranks, tree replicated, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # PMC-derived HBM bytes per launch (profiled at pairs_per_launch pairs; scaled
                         # linearly if this run uses another batch size)
                         "traffic": None if not traffic or not traffic.get("hbm_bytes_per_launch") else
                         traffic["hbm_bytes_per_launch"] * n / traffic.get("pairs_per_launch", n),
                         "kernel": "k_canopy_ilp" if info["strategy"] == "canopy" else "k_walk",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_pair": bytes_per_pair,
                         "mean_path_edges": h_mean, "pairs_per_launch": n,
                         "note": "achieved = algorithmic bytes of the reference's walk (28 + 8*h per pair, SURVEY 8d) / "
                                 "kernel time. The canopy kernel does not move those bytes (climb in LDS, understory "
                                 "pre-summed): frac > 1 is expected. Its real ceiling is the chip's random 64-B-sector "
                                 "read rate, see random_sector below and DESIGN.md section 5.2."},
            "kernel_pairs_per_s_per_gpu": n / (kernel_ms * 1e-3),
            "checksum": checksum,
        }
        if traffic and traffic.get("counters_mean_per_launch", {}).get("TCC_EA0_RDREQ_sum"):
            # profiled fabric read requests per pair x this run's pair rate, against the measured
            # ceiling for uniformly random 64-byte-sector reads (profiles/gather_microbench_r01.log)
            req_per_pair = traffic["counters_mean_per_launch"]["TCC_EA0_RDREQ_sum"] / traffic.get("pairs_per_launch", 1e8)
            rate = req_per_pair * n / (kernel_ms * 1e-3)
            line["random_sector"] = {"fabric_reads_per_pair": req_per_pair, "achieved_Greads_per_s": rate / 1e9,
                                     "ceiling_Greads_per_s": 59.0, "frac": rate / 59.0e9,
                                     "source": "rocprofv3 TCC_EA0_RDREQ_sum (profiles/) and scripts/micro/gather_bench.hip"}
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
        if world == 1 and not args.no_host_path:
            # end-to-end leg (SURVEY 8d asks for it next to the kernel-only figure; it is never
            # `value`): the same batch prefix from pageable host numpy arrays, through the
            # library's staged host path, into reused host result arrays
            k2 = min(n, 50_000_000)
            host_pairs = pairs[:k2].cpu().numpy()
            h_d, h_m = np.empty(k2), np.empty(k2, dtype=np.int32)
            tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
            t_h = time.perf_counter()
            tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
            t_h = time.perf_counter() - t_h
            line["end_to_end_host_path"] = {
                "pairs_per_s": k2 / t_h, "pairs": k2,
                "what": "pageable numpy int64 pairs in -> float64 distances + int32 MRCA ids out, PCIe inclusive "
                        "(ids cross as int32, distances as float32, widened on the host)",
                "matches_device_results": bool(np.array_equal(h_d.view(np.int64), out_d[:k2].cpu().numpy().view(np.int64))
                                               and np.array_equal(h_m, out_m[:k2].cpu().numpy()))}
        if world == 1 and not args.no_cpu_baseline:
            k = min(n, 50_000_000)
            cpu, parity = cpu_baseline(parent, dist, pairs[:k].cpu().numpy(), out_d[:k].cpu().numpy(),
                                       out_m[:k].cpu().numpy(), args.cpu_seconds)
            line["cpu_baseline"] = cpu
            line["parity"] = parity
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist_.barrier()
        dist_.destroy_process_group()
    tree.close()


def _depths(parent):
    from suchtree_amd.newick import node_depths
    return node_depths(parent).astype(np.int64)


if __name__ == "__main__":
    main()
