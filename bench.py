#!/usr/bin/env python3
"""Benchmark of the hot path: leaf-pair patristic distances (+ MRCA ids) per second
on BASELINE's 1M-leaf tree.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md section 8d "Config 3"): synthetic
perfectly balanced binary tree with 2^20 leaves (2,097,151 nodes, in-order ids),
ONE batch of 1e8 uniform random leaf pairs, int64 (n,2), resident in HBM on every
rank when the timed region starts.  One step = the whole batch: every rank computes one
contiguous slice on its own GPU (tree replicated, no data-path collective; rank 0's slice is
larger than the peers' so that its kernels end when their transfers do, --root-share) and
the result -- float64 distances + int32 MRCA ids for all n pairs --
is assembled on rank 0 by point-to-point RCCL transfers over xGMI, INSIDE the timed
region (suchtree_amd/sharding.py::run_sharded: float32 + int32 on the wire, sent in
pieces that overlap the next piece's kernel).  "scaling": "strong"; value = n pairs /
max-over-ranks step time.  With N = 1 a step is one kernel launch and nothing travels.
--weak restores per-rank batches without the gather (every rank its own --pairs pairs).

For N > 1 the driver launches this file with torch.distributed.run, one rank per GPU.
One JSON line on stdout from rank 0; see DESIGN.md section 6 for the fields.
"""
import argparse
import ctypes
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
    ap.add_argument("--pairs", type=int, default=100_000_000,
                    help="pairs per step (whole job; per rank with --weak)")
    ap.add_argument("--levels", type=int, default=20, help="balanced tree with 2**levels leaves")
    ap.add_argument("--strategy", default="auto", choices=["auto", "canopy", "walk"])
    ap.add_argument("--chunks", type=int, default=4, help="pieces per rank slice (transfer/compute overlap, N > 1)")
    ap.add_argument("--root-share", default="auto",
                    help="N > 1: fraction of the batch rank 0 computes itself: 'auto' (balance its kernels against the "
                         "peers' transfers, from rates measured before the timed region), 'even' (1/N) or a number")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank its own batch, no gather")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive end-to-end leg")
    ap.add_argument("--no-microbench", action="store_true", help="skip the in-process hardware-ceiling measurements")
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
        "sample": "first %d pairs of the batch, oracle/suchtree_oracle.c (visited-list MRCA, 20-byte AoS), "
                  "%d pthreads on contiguous chunks" % (n, cores),
        "single_thread_value": rate_1,
    }, {"distances_bit_exact": bit_exact, "mrca_bit_exact": mrca_exact, "max_rel_err": max_rel,
        "checked_pairs": n}


def sample_parity(parent, dist, pairs_t, out_d, out_m, plan):
    """N > 1: the assembled vector against the oracle around every slice and piece boundary."""
    from oracle.oracle import OracleTree
    O = OracleTree(parent, dist)
    idx = set()
    for g in range(plan.world):
        for lo, hi in plan.pieces(g):
            idx.update(range(max(0, lo - 500), min(plan.n, lo + 500)))
            idx.update(range(max(0, hi - 500), min(plan.n, hi)))
    idx = np.array(sorted(idx), dtype=np.int64)
    import torch
    it = torch.from_numpy(idx).to(pairs_t.device)
    p = pairs_t[it].cpu().numpy()
    d, m = out_d[it].cpu().numpy(), out_m[it].cpu().numpy()
    return {"distances_bit_exact": bool(np.array_equal(O.distances(p).view(np.int64), d.view(np.int64))),
            "mrca_bit_exact": bool(np.array_equal(O.mrca_bulk(p), m)), "checked_pairs": int(len(idx)),
            "what": "pairs within 500 of every slice / piece boundary of the assembled result, vs the oracle"}


def latest_traffic():
    """Per-launch counter summary of the rocprofv3 PMC passes of this same command, if one
    was committed (scripts/profile_gpu.sh -> profiles/traffic_rNN*.json)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), key=os.path.getmtime)
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def hardware_ceilings(device_index, footprint_bytes):
    """Measured in this process, on this GPU: the random 64-byte-sector read rate for a table
    the size of the record tables the kernel gathers from, and the streaming copy rate."""
    from suchtree_amd import build as st_build
    try:
        lib = ctypes.CDLL(st_build.MICRO_LIB)
    except OSError:
        return None     # helper library not built: the line simply carries no measured ceilings
    lib.stmb_random_sector_reads.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_double)]
    lib.stmb_stream_copy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    out = {}
    table = 1 << max(21, int(np.ceil(np.log2(max(footprint_bytes, 1)))))
    g = ctypes.c_double(0)
    for name, size in (("table", table), ("table_half", table // 2)):
        rc = lib.stmb_random_sector_reads(device_index, size, 32, 512, 3, ctypes.byref(g))
        if rc != 0:
            return None
        out[name] = {"MiB": size >> 20, "Greads_per_s": g.value}
    rc = lib.stmb_stream_copy(device_index, 1 << 30, 3, ctypes.byref(g))
    if rc == 0:
        out["stream_copy_GBps"] = g.value
    return out


def calibrate_root_share(tree, pairs, n, world, rank, device, stream, dist_, sharding):
    """Untimed, before the benchmark: this GPU's kernel rate on a prefix of the batch and the rate
    at which rank 0 receives from all peers at once (the gather's pattern), then the root's share
    of the batch for which its kernels and the peers' transfers end together
    (sharding.balanced_root_share).  Rank 0 decides; everyone gets its number."""
    import torch
    m = min(n, 20_000_000)
    d = torch.empty(m, dtype=torch.float32, device=device)
    mm = torch.empty(m, dtype=torch.int32, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(2):
        e0.record(stream)
        tree.distances_device(pairs.data_ptr(), m, d.data_ptr(), mm.data_ptr(), stream=stream.cuda_stream, f32=True)
        e1.record(stream)
    torch.cuda.synchronize(device)
    kernel_rate = m / (e0.elapsed_time(e1) * 1e-3)
    share, link, k_rate = sharding.measure_root_share(world, rank, kernel_rate, device=device)
    del d, mm
    return share, {"kernel_pairs_per_s": k_rate, "link_GBps_into_root_per_peer": link / 1e9}


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
        # a collective that involves every rank before the first batched point-to-point call
        # (torch.distributed.batch_isend_irecv: "if this is the first collective call in the
        # group ... all ranks must participate"): build the communicator here, deterministically
        dist_.barrier()

    from suchtree_amd import _capi, sharding, synth
    parent, dist = synth.balanced_tree(args.levels)
    n_leaves = 1 << args.levels
    tree = _capi.DeviceTree(parent, dist, device=local_rank, strategy=args.strategy)
    info = tree.info()

    # the synthetic batch, generated on the device (leaf ids are the even ids).  Strong scaling:
    # every rank generates the SAME batch (same seed) and owns a slice of it.
    n = args.pairs
    gen = torch.Generator(device=device)
    gen.manual_seed(3 + (rank if args.weak else 0))
    pairs = torch.randint(0, n_leaves, (n, 2), generator=gen, device=device, dtype=torch.int64) * 2
    stream = torch.cuda.current_stream(device)
    strong = not args.weak
    root_share, calib = None, None
    if strong and world > 1 and args.root_share != "even":
        if args.root_share == "auto":
            root_share, calib = calibrate_root_share(tree, pairs, n, world, rank, device, stream, dist_, sharding)
        else:
            root_share = float(args.root_share)
    plan = sharding.ShardPlan(n, world if strong else 1, rank if strong else 0,
                              chunks=args.chunks if (strong and world > 1) else 1, root_share=root_share)
    out_d, out_m, wire_d, wire_m = sharding.sharded_buffers(plan, device=device)
    piece_events = []

    def compute(lo, hi, dst_d, dst_m):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        tree.distances_device(pairs.data_ptr() + lo * 16, hi - lo, dst_d.data_ptr(), dst_m.data_ptr(),
                              stream=stream.cuda_stream, f32=dst_d.dtype == torch.float32)
        e1.record(stream)
        piece_events.append((e0, e1))

    def step():
        sharding.run_sharded(plan, compute, out_d, out_m, wire_d, wire_m)

    def barrier():
        if distributed:
            dist_.barrier()

    for _ in range(args.warmup):
        step()
    tree.fault_check(stream.cuda_stream)
    piece_events.clear()

    torch.cuda.synchronize(device)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = float(sum(a.elapsed_time(b) for a, b in piece_events)) / max(args.steps, 1)   # this rank, per step
    kernel_ms_max = kernel_ms
    if distributed:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
        dist_.all_reduce(t, op=dist_.ReduceOp.MAX)
        elapsed, kernel_ms_max = float(t[0].item()), float(t[1].item())
    tree.fault_check(stream.cuda_stream)
    lo0, hi0 = plan.bounds(plan.rank)
    pairs_this_rank = hi0 - lo0

    if rank == 0:
        # algorithmic bytes per pair: 16 in + 8 + 4 out + 8 per edge of the path,
        # h measured from the assembled results of the whole batch
        depth_t = torch.from_numpy(_depths(parent)).to(device)
        h = depth_t[pairs[:, 0]] + depth_t[pairs[:, 1]] - 2 * depth_t[out_m.long()]
        h_mean = float(h.double().mean().item())
        del h
        bytes_per_pair = 16 + 8 + 4 + 8 * h_mean
        checksum = float(out_d.sum().item())
        n_job = n * (world if args.weak else 1)
        value = float(n_job) * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        achieved = bytes_per_pair * pairs_this_rank / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_file = latest_traffic()
        kernel_name = {"canopy": "k_canopy_ilp", "walk": "k_walk"}.get(info["strategy"], info["strategy"])
        # bytes the algorithm has to request from the fabric per pair when no record is cache
        # resident: the coalesced streams plus one 64-byte sector per record read
        required = 16 + 12 + 2 * 64
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": kernel_name, "kernel_ms": kernel_ms, "pairs_per_launch": pairs_this_rank // plan.chunks,
                "launches_per_step": plan.chunks,
                "algorithmic_bytes_per_pair": bytes_per_pair, "mean_path_edges": h_mean,
                "required_bytes_per_pair": required,
                "required_GBps": required * pairs_this_rank / (kernel_ms * 1e-3) / 1e9,
                "required_frac": required * pairs_this_rank / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "achieved/frac: SURVEY 8d's algorithmic bytes of the reference's walk (28 + 8*h per pair) / "
                        "kernel time; the canopy kernel climbs in LDS and reads pre-summed understory records, so it "
                        "does not move those bytes and frac can exceed 1. required_*: the bytes this kernel must request "
                        "per pair (16 in + 12 out + two 64-byte record sectors). traffic: fabric bytes per launch from "
                        "request counts of the committed PMC passes (64 B per record request, 128 B per stream request, "
                        "+ WRITE_SIZE; calibration in profiles/README.md). The ceiling that binds is random_sector."}
        if traffic and traffic.get("hbm_bytes_per_launch"):
            roof["traffic"] = traffic["hbm_bytes_per_launch"] * (pairs_this_rank / plan.chunks) / traffic.get("pairs_per_launch", 1e8)
            roof["traffic_source"] = traffic_file
        line = {
            "metric": "leaf-pair patristic distances/sec (+ MRCA ids/sec), 1M-leaf tree",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "balanced 2^%d-leaf tree (%d nodes), %s, int64 ids in HBM -> float64 distance + "
                                   "int32 MRCA id%s"
                                   % (args.levels, len(parent),
                                      ("%d uniform random leaf pairs per GPU per step" % n) if args.weak else
                                      ("one batch of %d uniform random leaf pairs per step, sharded over %d GPU(s)" % (n, world)),
                                      "" if (args.weak or world == 1) else " assembled on rank 0 inside the timed region"),
                       "pairs_per_step": n_job, "pairs_per_gpu": pairs_this_rank if strong else n,
                       "tree_levels": args.levels, "kernel_family": info["strategy"],
                       "canopy_nodes": info["canopy_nodes"], "record_bytes": info["record_bytes"],
                       "sharding": ("contiguous pair slices (rank 0: %.0f %% of the batch, the peers share the rest: its kernels "
                                    "end when their transfers do), tree replicated, no data-path collective; results to rank 0 "
                                    "by RCCL send/recv over xGMI (float32 + int32 on the wire, %d pieces per slice)"
                                    % (100.0 * pairs_this_rank / n, plan.chunks))
                       if (strong and world > 1) else "none" if world == 1 else
                       "weak: every rank its own batch, tree replicated, nothing gathered"},
            "roofline": roof,
            "kernel_only_pairs_per_s": (n if strong else n * world) / (kernel_ms_max * 1e-3),
            "kernel_pairs_per_s_per_gpu": pairs_this_rank / (kernel_ms * 1e-3),
            "checksum": checksum,
        }
        if strong and world > 1:
            # what the gather costs on top of the slowest rank's kernels (exposed, after overlap)
            line["gather_ms"] = max(0.0, ms_per_step - kernel_ms_max)
            line["gather_bytes_into_root"] = 8 * (n - pairs_this_rank)
            line["root_share"] = pairs_this_rank / n
            if calib:
                line["root_share_calibration"] = calib
            line["parity"] = sample_parity(parent, dist, pairs, out_d, out_m, plan)
        if not args.no_microbench:
            # footprint the record gathers fall on: rec_a (8 B) + rec_b (record_bytes / 2) per leaf
            foot = n_leaves * (8 + info["record_bytes"] // 2) if info["strategy"] == "canopy" else len(parent) * 12
            hw = hardware_ceilings(local_rank, foot)
            if hw:
                line["hardware_measured"] = hw
                if hw.get("stream_copy_GBps"):
                    roof["measured_copy_GBps"] = hw["stream_copy_GBps"]
                    roof["frac_of_measured_copy"] = achieved / hw["stream_copy_GBps"]
                if traffic and traffic.get("counters_mean_per_launch", {}).get("TCC_EA0_RDREQ_sum"):
                    # fabric read requests per pair (committed PMC pass of this command) x this run's
                    # pair rate, against the random-sector rate measured a moment ago in this process
                    req_per_pair = traffic["counters_mean_per_launch"]["TCC_EA0_RDREQ_sum"] / traffic.get("pairs_per_launch", 1e8)
                    rate = req_per_pair * pairs_this_rank / (kernel_ms * 1e-3)
                    ceil = hw["table"]["Greads_per_s"]
                    line["random_sector"] = {
                        "fabric_reads_per_pair": req_per_pair, "achieved_Greads_per_s": rate / 1e9,
                        "ceiling_Greads_per_s": ceil, "ceiling_table_MiB": hw["table"]["MiB"],
                        "frac": rate / 1e9 / ceil,
                        "source": "requests: rocprofv3 TCC_EA0_RDREQ_sum in %s; ceiling: suchtree_amd/csrc/microbench.hip "
                                  "run in this process (random 32-byte reads, one per 64-byte sector)" % traffic_file}
        if world == 1:
            # MRCA ids alone (common_ancestors_bulk, quartets): on trees with in-order ids they come
            # from a rank table and a sparse table over the canopy, without the distance kernels
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            chk = torch.empty(n, dtype=torch.int32, device=device)
            tree.distances_device(pairs.data_ptr(), n, 0, chk.data_ptr(), stream=stream.cuda_stream)
            ev0.record(stream)
            for _ in range(5):
                tree.distances_device(pairs.data_ptr(), n, 0, chk.data_ptr(), stream=stream.cuda_stream)
            ev1.record(stream)
            ev1.synchronize()
            tree.fault_check(stream.cuda_stream)
            line["mrca_ids_only"] = {"ids_per_s": 5.0 * n / (ev0.elapsed_time(ev1) * 1e-3),
                                     "matches_the_fused_launch": bool(torch.equal(chk, out_m))}
            del chk
        if world == 1 and not args.no_host_path:
            # end-to-end leg (SURVEY 8d asks for it next to the kernel-only figure; it is never
            # `value`): the same batch prefix from pageable host numpy arrays through the library's
            # staged host path -- what T.distances_bulk(numpy) costs, PCIe inclusive
            k2 = min(n, 50_000_000)
            host_pairs = pairs[:k2].cpu().numpy()
            ref_d, ref_m = out_d[:k2].cpu().numpy(), out_m[:k2].cpu().numpy()
            h_d, h_m = np.empty(k2), np.empty(k2, dtype=np.int32)
            tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
            t_h = time.perf_counter()
            tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
            t_h = time.perf_counter() - t_h
            t_f = time.perf_counter()
            f_d, f_m = tree.distances_host(host_pairs, True, True)     # fresh result arrays, as the facade returns
            t_f = time.perf_counter() - t_f
            # what a caller's loop sees: "r = distances_host(...)", result dropped, again -- the
            # result blocks (>= 32 MiB) come back to the library's recycle pool and go out resident
            loop_ok = True
            t_l = time.perf_counter()
            for _ in range(3):
                l_d, l_m = tree.distances_host(host_pairs, True, True)
                loop_ok = loop_ok and bool(l_d[k2 - 1] == ref_d[k2 - 1] and l_m[0] == ref_m[0])
                del l_d, l_m
            t_l = (time.perf_counter() - t_l) / 3
            # opt-in: result arrays from the recycled pinned pool, written by the kernel directly
            tree.pinned_results = True
            p_d, p_m = tree.distances_host(host_pairs, True, True)
            pooled_ok = bool(np.array_equal(p_d.view(np.int64), ref_d.view(np.int64)) and np.array_equal(p_m, ref_m))
            del p_d, p_m
            t_p = time.perf_counter()
            p_d, p_m = tree.distances_host(host_pairs, True, True)
            t_p = time.perf_counter() - t_p
            del p_d, p_m
            tree.pinned_results = False
            line["end_to_end_host_path"] = {
                "pairs_per_s": k2 / t_h, "pairs_per_s_fresh_arrays": k2 / t_f,
                "pairs_per_s_call_and_drop_loop": k2 / t_l,
                "pairs_per_s_pinned_result_pool": k2 / t_p, "pairs": k2,
                "what": "pageable numpy int64 pairs in -> float64 distances + int32 MRCA ids out, PCIe inclusive "
                        "(ids cross as int32, distances as float32, widened on the host); reused result arrays / "
                        "result arrays allocated by the call, first use of their memory (what a single "
                        "SuchTree.distances_bulk call returns) / the same call in a loop that drops each result "
                        "(blocks recycled by the library, release included) / opt-in pinned result pool (float64 + "
                        "int32 written by the kernel straight into the returned arrays)",
                "matches_device_results": bool(np.array_equal(h_d.view(np.int64), ref_d.view(np.int64))
                                               and np.array_equal(h_m, ref_m)
                                               and np.array_equal(f_d.view(np.int64), ref_d.view(np.int64))
                                               and np.array_equal(f_m, ref_m) and pooled_ok and loop_ok)}
            del host_pairs, ref_d, ref_m, h_d, h_m, f_d, f_m
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
