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

How it starts.  `python bench.py --gpus N` with N > 1 and no torch.distributed.run
environment starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
CHILD process -- before this process has imported torch or loaded any HIP library -- relays
the child's JSON line, returns its exit code, and kills the child's process group when
--launch-timeout expires.  Launched by torch.distributed.run directly (the driver's form for
N > 1) the file is simply one rank.  One JSON line on stdout from rank 0; DESIGN.md section 6
explains the fields.  The CPU baseline (oracle, all host cores) and the parity block are
produced by rank 0 at every N, the peers waiting at the closing barrier.

Structure: `run_job` is backend-neutral (HipBackend here; the gloo test-suite injects a CPU
backend whose compute step is the oracle, tests/test_bench_gloo.py) -- the sharded step, the
barriers, the max-over-ranks timing, the deadline and the line are the same code on both.
"""
import argparse
import glob
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "leaf-pair patristic distances/sec (+ MRCA ids/sec), 1M-leaf tree"


def parse(argv=None):
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
    ap.add_argument("--gather", default="root", choices=["root", "allgather", "none"],
                    help="N > 1, strong scaling: where the result of a step ends up inside the timed region.  root (default): "
                         "assembled on rank 0 by point-to-point transfers; allgather: assembled on every rank (even slices, "
                         "grouped send/recv between all pairs of GPUs); none: every rank keeps its slice (nothing travels; the "
                         "slices are gathered once, untimed, for the parity check)")
    ap.add_argument("--no-gather-sweep", action="store_true",
                    help="N > 1, strong scaling: skip the few steps of the two OTHER gather modes timed after the timed region "
                         "(gather_modes in the line)")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="N > 1: CUs rank 0's kernels leave free for RCCL's receive kernels (handle option reserve_cus)")
    ap.add_argument("--wire-int32", action="store_true",
                    help="N > 1: MRCA ids travel as int32 (8 bytes per pair) instead of 24 bits each (7 bytes per pair)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive end-to-end leg")
    ap.add_argument("--no-microbench", action="store_true", help="skip the in-process hardware-ceiling measurements")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs on BASELINE configs 2, 4 and 5 (other_configs in the line)")
    ap.add_argument("--deadline", type=float, default=600.0,
                    help="seconds every rank allows for warmup + timed steps; a rank that misses it exits with code 3")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="N > 1 self-launch: seconds before the parent kills the torch.distributed.run child")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------
# self-launch (N > 1 without a torch.distributed.run environment)
# --------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(n_gpus, argv, port):
    """The command the driver itself uses for N > 1 (task contract), as a child of this process.
    SUCHTREE_AMD_BENCH_LAUNCHER replaces the launcher module (the CPU test-suite's stand-in)."""
    module = os.environ.get("SUCHTREE_AMD_BENCH_LAUNCHER", "torch.distributed.run")
    head = [sys.executable, module] if module.endswith(".py") else [sys.executable, "-m", module]
    return head + ["--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """Parent side of `python bench.py --gpus N` (N > 1).  Nothing here may touch the GPU: no torch,
    no HIP library (the box forbids replacing or forking a process that has initialised the GPU, so
    the ranks are children of a process that never did)."""
    # the ranks all load libsuchtree_hip.so at once: make sure it is built before they start (hipcc needs no
    # GPU; suchtree_amd.build imports neither torch nor the library), so that no rank has to build it
    try:
        from suchtree_amd import build as st_build
        st_build.build()
        st_build.build_microbench()
    except Exception as e:      # noqa: BLE001 -- the ranks will report a missing library themselves
        sys.stderr.write("bench.py: building the library in the launcher's parent failed (%s)\n" % e)
    cmd = launcher_command(args.gpus, argv, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    try:
        out, _ = child.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        child.wait()
        sys.stderr.write("bench.py: the %d-rank job did not finish within %.0f s; killed\n" % (args.gpus, args.launch_timeout))
        return 124
    lines = [l for l in out.decode("utf-8", "replace").splitlines() if l.lstrip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if child.returncode != 0:
        sys.stderr.write("bench.py: torch.distributed.run exited with code %d\n" % child.returncode)
        return child.returncode if child.returncode > 0 else 1
    if not lines:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no JSON line\n")
        return 1
    return 0


class Deadline:
    """All-ranks deadline: every rank arms one around its warmup + timed steps; a rank still inside
    when it expires (a hung peer, a lost link) says so and exits with code 3, which makes
    torch.distributed.run tear the job down instead of waiting for RCCL's own 10-minute timeout."""

    def __init__(self, seconds, what, rank=0, on_expire=None):
        self.seconds, self.what, self.rank = float(seconds), what, rank
        self.on_expire = on_expire or self._die
        self._timer = None

    def _die(self):
        sys.stderr.write("bench.py: rank %d missed the %.0f s deadline for %s; exiting\n" % (self.rank, self.seconds, self.what))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            self._timer = threading.Timer(self.seconds, self.on_expire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


# --------------------------------------------------------------------------------------------
# CPU legs (oracle = checker and baseline; never part of the timed GPU region)
# --------------------------------------------------------------------------------------------
def cpu_baseline(parent, dist, pairs_host, gpu_dist, gpu_mrca, seconds):
    """Oracle (CPU port of the reference algorithm) on a bounded sample of the same
    pairs, all host cores, fork-pool-style contiguous chunks.  Also the parity check
    of the timed GPU results on that sample."""
    from oracle import oracle as orc
    O = orc.OracleTree(parent, dist)
    port = O
    kind, what = "port", "oracle/suchtree_oracle.c (visited-list MRCA, 20-byte AoS)"
    if orc.ref_lib() is not None:
        # the reference's OWN compiled _distances / _mrca (oracle/_ref/libref_hotpath.so: SuchTree/MuchTree.c as shipped, through
        # oracle/ref_harness.c) is the timed baseline and the parity checker wherever the file travelled with the snapshot
        O = orc.RefTree(parent, dist, depth=port.depth)
        kind, what = "reference", "the reference's compiled SuchTree._distances (SuchTree/MuchTree.c as shipped, oracle/ref_harness.c)"
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
        "value": n / dt, "unit": "pairs/s", "cores": cores, "kind": kind,
        # cores = the threads used = this process's affinity mask (a launcher may narrow it per rank); the host has host_cpus
        "host_cpus": os.cpu_count(),
        "sample": "first %d pairs of the batch, %s, %d pthreads on contiguous chunks" % (n, what, cores),
        "single_thread_value": rate_1,
    }, {"distances_bit_exact": bit_exact, "mrca_bit_exact": mrca_exact, "max_rel_err": max_rel,
        "checked_pairs": n, "checked_against": kind,
        # (the restatement itself against the reference's code on a slice of the same pairs, where both are there)
        "oracle_equals_reference": bool(np.array_equal(port.distances(pairs_host[:200_000]).view(np.int64),
                                                       O.distances(pairs_host[:200_000]).view(np.int64))) if kind == "reference" else None}


def spread_parity(parent, dist, pairs_t, out_d, out_m, plan, per_edge=500, strided=1_000_000):
    """The assembled result against the oracle (a) around every slice and piece boundary and (b) on
    an evenly strided sample of the whole batch, so that every rank's slice is covered."""
    import torch
    from oracle.oracle import OracleTree
    O = OracleTree(parent, dist)
    idx = set()
    if plan.world > 1:
        for g in range(plan.world):
            for lo, hi in plan.pieces(g):
                idx.update(range(max(0, lo - per_edge), min(plan.n, lo + per_edge)))
                idx.update(range(max(0, hi - per_edge), min(plan.n, hi)))
    n_edge = len(idx)
    if plan.n:
        idx.update(range(0, plan.n, max(1, plan.n // max(1, strided))))
    idx = np.array(sorted(idx), dtype=np.int64)
    it = torch.from_numpy(idx).to(pairs_t.device)
    p = pairs_t[it].cpu().numpy()
    d, m = out_d[it].cpu().numpy(), out_m[it].cpu().numpy()
    cores = len(os.sched_getaffinity(0))
    return {"distances_bit_exact": bool(np.array_equal(O.distances_mt(p, cores).view(np.int64), d.view(np.int64))),
            "mrca_bit_exact": bool(np.array_equal(O.mrca_bulk(p), m)), "checked_pairs": int(len(idx)),
            "what": "assembled result vs the oracle: %d pairs within %d of every slice / piece boundary + every %d-th "
                    "pair of the batch" % (n_edge, per_edge, max(1, plan.n // max(1, strided)))}


def latest_traffic():
    """Per-launch counter summary of the rocprofv3 PMC passes of this same command, if one
    was committed (scripts/profile_gpu.sh -> profiles/traffic_rNN*.json)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    files = [f for f in files if "_ml_" not in f and "_walk_" not in f and "_tri_" not in f]
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def footprint_sweep():
    """The committed footprint sweep of the same kernel family (scripts/footprint_sweep.sh -> profiles/footprint_sweep_rNN.json):
    balanced trees of 2^20 / 2^22 / 2^24 leaves, whose record tables go from inside the Infinity Cache to four times its size."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "footprint_sweep_r*.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:      # noqa: BLE001
        return None
    keys = ("leaves", "gather_footprint_bytes", "fits_infinity_cache", "avg_ns", "pairs_per_s", "counter_bytes_per_pair",
            "counter_TBps", "frac_of_hbm_peak", "fabric_read_requests_per_pair", "l2_hit_rate")
    return {"source": os.path.relpath(files[-1], ROOT), "trees": [{k: t.get(k) for k in keys} for t in d.get("trees", [])],
            "what": "rocprofv3 kernel time and counter bytes of st_distances_device, 1e8 random leaf pairs per launch; beyond 256 MiB "
                    "of gathered records the counter bytes are HBM traffic"}


# --------------------------------------------------------------------------------------------
# the MI355X backend
# --------------------------------------------------------------------------------------------
class HipBackend:
    """The product path: suchtree_amd._capi.DeviceTree on this rank's GPU, torch only for device
    buffers, events and the process group."""

    name = "hip"

    def __init__(self, args, parent, dist, local_rank):
        import torch
        from suchtree_amd import _capi
        self.torch = torch
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.device = torch.device("cuda", local_rank)
        self.stream = torch.cuda.current_stream(self.device)
        self.tree = _capi.DeviceTree(parent, dist, device=local_rank, strategy=args.strategy)
        if args.reserve_cus and int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0:
            self.tree.set_option("reserve_cus", args.reserve_cus)
        self.n_leaves = 1 << args.levels
        self._events = []

    def info(self):
        return self.tree.info()

    def make_pairs(self, n, seed):
        """SURVEY 8d config 3's batch: default_rng(seed).integers(0, leaves, (n, 2)) * 2 (leaf ids), int64, drawn on the
        host and uploaded once, before anything is timed (seed 3: the batch of the -m gpu suite's headline-launch test)."""
        from suchtree_amd import synth
        return self.torch.from_numpy(synth.random_leaf_pairs(self.n_leaves, n, seed=seed)).to(self.device)

    def bind(self, pairs):
        torch, tree, stream = self.torch, self.tree, self.stream

        def compute(lo, hi, dst_d, dst_m):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if dst_m.dtype == torch.uint8:      # the gather's wire format: float32 + 24-bit ids, packed by the kernel itself
                tree.distances_device_wire(pairs.data_ptr() + lo * 16, hi - lo, dst_d.data_ptr(), dst_m.data_ptr(),
                                           stream=stream.cuda_stream)
            else:
                tree.distances_device(pairs.data_ptr() + lo * 16, hi - lo, dst_d.data_ptr(), dst_m.data_ptr(),
                                      stream=stream.cuda_stream, f32=dst_d.dtype == torch.float32)
            e1.record(stream)
            self._events.append((e0, e1))
        return compute

    def unpack_mrca24(self, packed, out):
        """Root side of the packed wire format: the library's kernel on the current stream (sharding.run_sharded)."""
        self.tree.unpack_mrca24_device(packed.data_ptr(), out.numel(), out.data_ptr(), stream=self.stream.cuda_stream)

    def kernel_clock_reset(self):
        self._events.clear()

    def kernel_ms_total(self):
        return float(sum(a.elapsed_time(b) for a, b in self._events))

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def fault_check(self):
        self.tree.fault_check(self.stream.cuda_stream)

    def kernel_rate(self, pairs, m):
        """This GPU's kernel rate on a prefix of the batch (for the root-share calibration)."""
        torch = self.torch
        d = torch.empty(m, dtype=torch.float32, device=self.device)
        mm = torch.empty(m, dtype=torch.int32, device=self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            e0.record(self.stream)
            self.tree.distances_device(pairs.data_ptr(), m, d.data_ptr(), mm.data_ptr(), stream=self.stream.cuda_stream, f32=True)
            e1.record(self.stream)
        self.synchronize()
        return m / (e0.elapsed_time(e1) * 1e-3)

    def close(self):
        self.tree.close()

    def extra_legs(self, args, line, roof, pairs, out_d, out_m, traffic, traffic_file, pairs_this_rank, kernel_ms):
        """Rank 0 only, after the timed region: hardware ceilings, MRCA-only launch, host path, other configs."""
        import bench_legs
        info = self.info()
        n = pairs.shape[0]
        if not args.no_microbench:
            # the tables the kernel gathers from, for leaf pairs: rec_b (half a record per leaf) and rec_a4 (4 B) or
            # rec_a (8 B); walk family: node records + lineage blocks are not modelled (12 B per node as a floor)
            foot = gather_footprint_bytes(info, self.n_leaves)
            hw = bench_legs.hardware_ceilings(self.local_rank, foot)
            if hw:
                line["hardware_measured"] = hw
                pairs_per_s = pairs_this_rank / (kernel_ms * 1e-3)
                three = bench_legs.three_ceilings(traffic, pairs_per_s, roof["algorithmic_bytes_per_pair"], hw.get("table"),
                                                  hw.get("stream_copy_GBps"))
                roof.update({k: three[k] for k in ("counter_traffic", "request_rate") if k in three})
                if hw.get("stream_copy_GBps"):
                    roof["measured_copy_GBps"] = hw["stream_copy_GBps"]
                rr = three.get("request_rate", {})
                if rr.get("frac"):
                    # what limits a gather kernel below the byte peak: the fabric's random-sector request rate at this footprint
                    # (measured here, in this process; the round's full sweep is committed as profiles/ceilings_sweep_rNN.json)
                    roof["secondary_ceiling"] = {
                        "name": "fabric_random_sector", "achieved": rr["Greads_per_s"], "peak": rr["ceiling_Greads_per_s"],
                        "unit": "Greads/s", "frac": rr["frac"], "gather_footprint_MiB": foot / 2**20,
                        "committed_sweep": bench_legs.committed_sector_ceiling(foot),
                        "ceiling_source": "suchtree_amd/csrc/microbench.hip run in this process: random 32-byte reads, one per 64-byte "
                                          "sector, from a table of the kernel's own gather footprint; best of a sweep over unroll and "
                                          "grid shape; requests per pair: rocprofv3 TCC_EA0_RDREQ_sum in %s" % traffic_file}
                    # The same comparison in the unit the CU's miss path counts in (round 6, profiles/counters_headline_r06.txt: TA busy 93 %
                    # of the launch, the L1 stalled on pending misses 85 %): L1-miss requests per second -- fabric reads PLUS the L2 hits
                    # (rec_a4 entries, a quarter of the kernel's misses), which occupy a miss slot without becoming a fabric read -- against
                    # the microbenchmark's lane reads per second, each of which is one L1-miss request (ceiling_counters_rNN.json).
                    l1 = (traffic or {}).get("counters_mean_per_launch", {}).get("TCP_TCC_READ_REQ_sum")
                    if l1 and traffic.get("pairs_per_launch"):
                        per_pair = l1 / traffic["pairs_per_launch"]
                        cc = bench_legs.committed_ceiling_counters()
                        roof["secondary_ceiling"]["l1_miss_requests"] = {
                            "per_pair": per_pair, "Greq_per_s": per_pair * pairs_per_s / 1e9,
                            "frac_of_ceiling_lane_reads": per_pair * pairs_per_s / 1e9 / rr["ceiling_Greads_per_s"],
                            "ceiling_l1_miss_requests_per_lane_read": (cc or {}).get("l1_per_read"),
                            "ceiling_fabric_requests_per_lane_read": (cc or {}).get("fabric_per_read"),
                            "ceiling_counters_from": (cc or {}).get("source"),
                            "why": "the microbenchmark's G reads/s are lane reads = L1-miss requests; the kernel's request_rate counts fabric "
                                   "reads only, so `frac` understates how close the CU's miss path is to its ceiling"}
        line["mrca_ids_only"] = bench_legs.mrca_ids_only(self, pairs, out_m)
        if not args.no_host_path:
            line["end_to_end_host_path"] = bench_legs.host_path_leg(self, pairs, out_d, out_m)
        if not args.no_other_configs:
            line["other_configs"] = bench_legs.other_configs(self)


# --------------------------------------------------------------------------------------------
# the job (backend-neutral)
# --------------------------------------------------------------------------------------------
def _depths(parent):
    from suchtree_amd.newick import node_depths
    return node_depths(parent).astype(np.int64)


def run_job(args, be, dg, world, rank, parent, dist, peers_wait=None):
    """Warmup, the timed sharded steps, the line.  `be`: backend (HipBackend, or the test-suite's CPU
    backend); `dg`: torch.distributed with an initialised default group, or None (plain one-process
    run); `peers_wait`: what ranks > 0 do after the timed region while rank 0 finishes the line.
    Returns the line on rank 0, None elsewhere."""
    import torch
    from suchtree_amd import sharding

    info = be.info()
    n = args.pairs
    strong = not args.weak
    # strong scaling: every rank generates the SAME batch (same seed) and owns a slice of it
    pairs = be.make_pairs(n, 3 + (rank if args.weak else 0))
    root_share, calib = None, None
    sweep = strong and world > 1 and not args.no_gather_sweep
    if strong and world > 1 and args.root_share != "even" and (args.gather == "root" or sweep):
        if args.root_share == "auto":
            # untimed: this GPU's kernel rate on a prefix of the batch and the rate at which rank 0
            # receives from all peers at once (the gather's pattern); rank 0 decides, everyone agrees
            k_rate = be.kernel_rate(pairs, min(n, 20_000_000))
            root_share, link, k_rate = sharding.measure_root_share(
                world, rank, k_rate, device=be.device, nbytes=getattr(be, "calibration_bytes", 64 << 20),
                wire_bytes_per_pair=sharding.WIRE_BYTES_PLAIN if (args.wire_int32 or info["n_nodes"] > 0xFFFFFF) else sharding.WIRE_BYTES_PACKED)
            wb = sharding.WIRE_BYTES_PLAIN if (args.wire_int32 or info["n_nodes"] > 0xFFFFFF) else sharding.WIRE_BYTES_PACKED
            calib = {"kernel_pairs_per_s": k_rate, "link_GBps_into_root_per_peer": link / 1e9, "root_share": root_share,
                     # DESIGN.md section 7's projection (60 GB/s per link into the root, this GPU's measured kernel rate) beside it
                     "projected_root_share_at_60GBps_per_link": sharding.balanced_root_share(world, k_rate, 60e9, wb),
                     "projected_speedup_at_60GBps_per_link": 1.0 / max(sharding.balanced_root_share(world, k_rate, 60e9, wb), 1e-9),
                     "calibrated_speedup": 1.0 / max(root_share, 1e-9),
                     "what": "untimed, before the timed region: this GPU's kernel rate on a 2e7-pair prefix and the rate at which rank 0 "
                             "receives from all peers at once; root_share = the fraction of the batch for which rank 0's kernels end when "
                             "the peers' transfers do; 1 / root_share bounds the root gather's speedup over one GPU"}
        else:
            root_share = float(args.root_share)
    # the gather's wire format: float32 + 24-bit MRCA id (7 bytes per pair) on trees of fewer than 2^24 nodes, packed by
    # the peers' kernels, unpacked on the root piece by piece; else float32 + int32
    mode = args.gather if (strong and world > 1) else "root"
    packed = strong and world > 1 and info["n_nodes"] <= 0xFFFFFF and not args.wire_int32
    wire_bytes = sharding.WIRE_BYTES_PACKED if packed else sharding.WIRE_BYTES_PLAIN
    plan = sharding.ShardPlan(n, world if strong else 1, rank if strong else 0,
                              chunks=args.chunks if (strong and world > 1) else 1,
                              root_share=root_share if mode == "root" else None, align=4)
    out_d, out_m, wire_d, wire_m = sharding.sharded_buffers(plan, device=be.device, packed_ids=packed, all_ranks=mode == "allgather")
    compute = be.bind(pairs)
    unpack = getattr(be, "unpack_mrca24", None)
    own_d = own_m = None
    if mode == "none":      # every rank keeps its slice: float64 + int32 of the slice's length, written by the kernels directly
        s_lo, s_hi = plan.bounds(plan.rank)
        own_d = torch.empty(s_hi - s_lo, dtype=torch.float64, device=be.device)
        own_m = torch.empty(s_hi - s_lo, dtype=torch.int32, device=be.device)

    def step():
        if mode == "allgather":
            sharding.run_allgather(plan, compute, out_d, out_m, wire_d, wire_m, unpack=unpack)
        elif mode == "none":
            sharding.run_local(plan, compute, own_d, own_m)
        else:
            sharding.run_sharded(plan, compute, out_d, out_m, wire_d, wire_m, unpack=unpack)

    def barrier():
        if dg is not None:
            dg.barrier()

    with Deadline(args.deadline, "warmup + %d timed steps" % args.steps, rank):
        for _ in range(args.warmup):
            step()
        be.fault_check()
        be.kernel_clock_reset()

        be.synchronize()
        barrier()
        be.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        be.synchronize()
        barrier()
        t1 = time.perf_counter()
        elapsed = t1 - t0
        kernel_ms = be.kernel_ms_total() / max(args.steps, 1)      # this rank, per step
        kernel_ms_max = kernel_ms
        if dg is not None:
            t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=be.device)
            dg.all_reduce(t, op=dg.ReduceOp.MAX)
            elapsed, kernel_ms_max = float(t[0].item()), float(t[1].item())
        be.fault_check()
    lo0, hi0 = plan.bounds(plan.rank)
    pairs_this_rank = hi0 - lo0
    per_rank = None
    if dg is not None:
        # what every rank did, as seen on its own GPU: kernel milliseconds per step and pairs per step (the driver's first
        # multi-GPU run should show whether the kernels scale even where the gather does not)
        mine = torch.tensor([kernel_ms, float(pairs_this_rank)], dtype=torch.float64, device=be.device)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dg.all_gather(everyone, mine)
        per_rank = {"kernel_ms": [float(t[0].item()) for t in everyone], "pairs": [int(t[1].item()) for t in everyone]}
    if mode == "none":
        # untimed: the slices are assembled on rank 0 once (the root gather with the same plan), so that the parity
        # check and the path statistics below see the whole batch (kernel_ms was read above)
        out_d, out_m, wire_d, wire_m = sharding.sharded_buffers(plan, device=be.device, packed_ids=packed)
        sharding.run_sharded(plan, compute, out_d, out_m, wire_d, wire_m, unpack=unpack)
        be.synchronize()
        if rank == 0:      # the timed slice of this rank must be what the assembled result holds
            if not (torch.equal(own_d.view(torch.int64), out_d[lo0:hi0].view(torch.int64)) and torch.equal(own_m, out_m[lo0:hi0])):
                raise SystemExit("bench.py: --gather none: rank 0's resident slice differs from the assembled result")
        be.fault_check()

    gather_modes = None
    if sweep:
        # After the timed region, a few steps of each of the other two gather modes, timed the same way (barrier, sync, max
        # over ranks), so that ONE run at N GPUs shows what the gather costs: kernels alone (none), every link at once
        # (allgather), everything into rank 0 (root).  Not part of `value`.
        gather_modes = {mode: {"ms_per_step": elapsed / args.steps * 1e3, "pairs_per_s": n * args.steps / elapsed, "steps": args.steps,
                               "timed_region": True, "kernel_ms_slowest_rank": kernel_ms_max,
                               "gather_ms": max(0.0, elapsed / args.steps * 1e3 - kernel_ms_max),
                               "root_share": (hi0 - lo0) / max(n, 1) if rank == 0 else None}}
        k = max(1, min(args.steps, 5))
        for other in ("root", "allgather", "none"):
            if other == mode:
                continue
            with Deadline(args.deadline, "the %s gather mode's %d steps" % (other, k), rank):
                o_plan = sharding.ShardPlan(n, world, rank, chunks=args.chunks, root_share=root_share if other == "root" else None, align=4)
                o_d, o_m, o_wd, o_wm = sharding.sharded_buffers(o_plan, device=be.device, packed_ids=packed, all_ranks=other == "allgather")
                o_lo, o_hi = o_plan.bounds(rank)
                o_own = (torch.empty(o_hi - o_lo, dtype=torch.float64, device=be.device),
                         torch.empty(o_hi - o_lo, dtype=torch.int32, device=be.device)) if other == "none" else None

                def o_step():
                    if other == "allgather":
                        sharding.run_allgather(o_plan, compute, o_d, o_m, o_wd, o_wm, unpack=unpack)
                    elif other == "none":
                        sharding.run_local(o_plan, compute, o_own[0], o_own[1])
                    else:
                        sharding.run_sharded(o_plan, compute, o_d, o_m, o_wd, o_wm, unpack=unpack)

                o_step()
                be.synchronize()
                be.kernel_clock_reset()
                barrier()
                be.synchronize()
                t0 = time.perf_counter()
                for _ in range(k):
                    o_step()
                be.synchronize()
                barrier()
                dt = torch.tensor([time.perf_counter() - t0, be.kernel_ms_total() / k], dtype=torch.float64, device=be.device)
                dg.all_reduce(dt, op=dg.ReduceOp.MAX)
                be.fault_check()
                o_ms = float(dt[0].item()) / k * 1e3
                # kernel_ms_slowest_rank: kernels alone (HIP events on the launch stream, max over ranks); gather_ms: what the
                # mode's transfers add on top after overlap -- one run per N separates kernel scaling from link limits
                gather_modes[other] = {"ms_per_step": o_ms, "pairs_per_s": n * k / float(dt[0].item()), "steps": k,
                                       "timed_region": False, "kernel_ms_slowest_rank": float(dt[1].item()),
                                       "gather_ms": max(0.0, o_ms - float(dt[1].item())),
                                       "root_share": (o_hi - o_lo) / max(n, 1) if rank == 0 else None}
                del o_d, o_m, o_wd, o_wm, o_own

    line = None
    if rank != 0 and peers_wait is not None:
        peers_wait()
    if rank == 0:
        # algorithmic bytes per pair: 16 in + 8 + 4 out + 8 per edge of the path,
        # h measured from the assembled results of the whole batch
        depth_t = torch.from_numpy(_depths(parent)).to(be.device)
        h = depth_t[pairs[:, 0]] + depth_t[pairs[:, 1]] - 2 * depth_t[out_m.long()]
        h_mean = float(h.double().mean().item())
        del h
        checksum = float(out_d.sum().item())
        traffic, traffic_file = latest_traffic()
        # the roofline block speaks for the SLOWEST rank's kernels (the rank that bounds the step), not for rank 0's own
        slow = None
        if per_rank and world > 1:
            r = int(np.argmax(per_rank["kernel_ms"]))
            slow = {"rank": r, "kernel_ms": per_rank["kernel_ms"][r], "pairs": per_rank["pairs"][r]}
        line = build_line(args, world, plan, info, len(parent), elapsed, kernel_ms, kernel_ms_max, h_mean, checksum,
                          pairs_this_rank, calib, traffic, traffic_file, wire_bytes, mode, slow=slow)
        if dg is not None:
            line["process_group"] = {"backend": str(dg.get_backend()), "world_size": int(dg.get_world_size()),
                                     "what": "as torch.distributed reports them for the group the step ran on (nccl = RCCL on ROCm)"}
        if per_rank:
            line["per_rank"] = per_rank
        if gather_modes:
            line["gather_modes"] = gather_modes
        roof = line["roofline"]
        if "traffic_not_used" in roof:      # (the side figures must not borrow another configuration's counters either)
            traffic = None
        if hasattr(be, "extra_legs"):
            be.extra_legs(args, line, roof, pairs, out_d, out_m, traffic, traffic_file, pairs_this_rank, kernel_ms)
        if strong and world > 1:
            line["parity_across_slices"] = spread_parity(parent, dist, pairs, out_d, out_m, plan)
        if not args.no_cpu_baseline:
            k = min(n, 50_000_000)
            cpu, parity = cpu_baseline(parent, dist, pairs[:k].cpu().numpy(), out_d[:k].cpu().numpy(),
                                       out_m[:k].cpu().numpy(), args.cpu_seconds)
            line["cpu_baseline"] = cpu
            line["parity"] = parity
            if strong and world > 1:
                # at N > 1 the first pairs all belong to rank 0's slice: the verdict on the whole
                # assembled vector needs the spread sample too
                s = line["parity_across_slices"]
                parity["distances_bit_exact"] = parity["distances_bit_exact"] and s["distances_bit_exact"]
                parity["mrca_bit_exact"] = parity["mrca_bit_exact"] and s["mrca_bit_exact"]
                parity["checked_pairs"] += s["checked_pairs"]
    return line


def gather_footprint_bytes(info, n_leaves):
    """The tables the kernel gathers from, for leaf pairs: rec_b or the cherry records (half a record per leaf, or a quarter) and
    rec_a4 (4 B) or rec_a (8 B); walk family: node records + lineage blocks are not modelled (12 B per node as a floor)."""
    a_bytes = info.get("a_side_bytes") or 8
    b_bytes = info.get("b_table_bytes_per_leaf") or info["record_bytes"] // 2
    return n_leaves * (a_bytes + b_bytes) if info["strategy"] == "canopy" else info["n_nodes"] * 12


def traffic_speaks_for(traffic, args, info, world):
    """A committed PMC summary is used for `roofline.achieved` only when it was taken on THIS configuration: the same tree
    (levels, canopy, record size) and the same kernel.  Summaries since round 6 carry `config` (scripts/profile_gpu.sh);
    older ones are accepted for the default tree only."""
    if not traffic or not traffic.get("hbm_bytes_per_launch") or not traffic.get("pairs_per_launch"):
        return False, "no committed PMC summary"
    if info["strategy"] != "canopy" or "k_canopy_ilp" not in str(traffic.get("kernel_full_name", "")):
        return False, "the committed summary is of %s, this run's kernel family is %s" % (traffic.get("kernel_full_name"), info["strategy"])
    c = traffic.get("config")
    if c is None:
        ok = args.levels == 20
        return ok, None if ok else "the committed summary carries no config and this is not the default tree"
    for key, mine in (("levels", args.levels), ("canopy_nodes", info["canopy_nodes"]), ("record_bytes", info["record_bytes"])):
        if key in c and int(c[key]) != int(mine):
            return False, "the committed summary was taken with %s = %s, this run has %s" % (key, c[key], mine)
    return True, None


def build_line(args, world, plan, info, n_nodes, elapsed, kernel_ms, kernel_ms_max, h_mean, checksum,
               pairs_this_rank, calib, traffic, traffic_file, wire_bytes=8, mode="root", slow=None):
    """The contract's JSON line from the measured quantities (no measurement happens here).  `slow`: at N > 1 the
    slowest rank's {rank, kernel_ms, pairs}: the roofline block is computed from ITS kernel time."""
    n = args.pairs
    roof_pairs, roof_ms, roof_rank = pairs_this_rank, kernel_ms, 0
    if slow and slow.get("kernel_ms") and slow.get("pairs"):
        roof_pairs, roof_ms, roof_rank = int(slow["pairs"]), float(slow["kernel_ms"]), int(slow["rank"])
    strong = not args.weak
    bytes_per_pair = 16 + 8 + 4 + 8 * h_mean
    n_job = n * (world if args.weak else 1)
    value = float(n_job) * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    achieved = bytes_per_pair * roof_pairs / (roof_ms * 1e-3) / 1e9
    kernel_name = {"canopy": "k_canopy_ilp", "walk": "k_walk"}.get(info["strategy"], info["strategy"])
    rate = roof_pairs / (roof_ms * 1e-3)      # the (slowest) rank's pairs per second of kernel time (HIP events on the launch stream)
    # bytes the kernel has to request from the fabric per pair when no record is cache resident: the coalesced
    # streams plus one 64-byte sector per record read
    required = 16 + 12 + 2 * 64
    # The block follows SURVEY 8d with HBM as the bound.  Primary figure: the bytes the kernel MOVES -- fabric bytes per
    # launch from the committed rocprofv3 PMC passes of this same command (profiles/traffic_rNN.json; per pair, scaled to
    # this launch) over this run's kernel time.  SURVEY 8d's algorithmic bytes (28 + 8 h per pair: the reference's walk)
    # stay beside it in `algorithmic`: the canopy kernel climbs in LDS and reads pre-summed understories, so it does not
    # move them and that fraction exceeds 1 -- it is not a bandwidth claim.
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None,
            "kernel": kernel_name, "kernel_ms": roof_ms, "pairs_per_launch": roof_pairs // plan.chunks,
            "launches_per_step": plan.chunks}
    if world > 1:
        roof["rank"] = roof_rank
        roof["rank_is"] = "the rank with the longest kernel time per step (it bounds the step); kernel_ms and pairs are its own"
    per_pair = None
    speaks, why_not = traffic_speaks_for(traffic, args, info, world)
    if not speaks:
        roof["traffic_not_used"] = why_not
    if speaks:
        per_pair = traffic["hbm_bytes_per_launch"] / traffic["pairs_per_launch"]
        roof["traffic"] = per_pair * (roof_pairs / plan.chunks)
        roof["traffic_bytes_per_pair"] = per_pair
        roof["traffic_source"] = traffic_file
        c = traffic.get("counters_mean_per_launch", {})
        if c.get("TCC_HIT_sum") and c.get("TCC_MISS_sum"):
            roof["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])       # same PMC passes
        k = (traffic.get("kernels") or {}).get(traffic.get("kernel_full_name"), {})
        if k.get("avg_ns"):
            # the same quotient from the committed files alone: counter bytes per launch / rocprofv3's average duration
            roof["rocprof"] = {"kernel": traffic.get("kernel_full_name"), "calls": k.get("calls"), "kernel_avg_ms": k["avg_ns"] * 1e-6,
                               "counter_GBps": traffic["hbm_bytes_per_launch"] / (k["avg_ns"] * 1e-9) / 1e9,
                               "frac": traffic["hbm_bytes_per_launch"] / (k["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS,
                               "source": "%s (rocprofv3 --kernel-trace --stats + one --pmc pass per counter set of `python3 bench.py "
                                         "--steps 5 --warmup 2`; kernel_stats_%s.csv is the stats table)"
                                         % (traffic_file, str(traffic.get("tag")))}
    moved = per_pair if per_pair is not None else float(required)
    roof["achieved"] = moved * rate / 1e9
    roof["frac"] = roof["achieved"] / HBM_PEAK_GBPS
    roof["achieved_is"] = ("fabric bytes per pair by counters (%s: read requests x calibrated bytes per request + WRITE_SIZE) x this "
                           "run's pairs per second of kernel time" % traffic_file) if per_pair is not None else \
                          ("no committed PMC pass of this configuration (%s): the bytes the kernel must request per pair (16 in + 12 out + "
                           "two 64-byte record sectors)" % why_not)
    alg_frac = achieved / HBM_PEAK_GBPS
    roof["algorithmic"] = {"bytes_per_pair": bytes_per_pair, "mean_path_edges": h_mean, "GBps": achieved, "frac_of_hbm_peak": alg_frac,
                           "exceeds_peak": bool(alg_frac > 1.0),
                           "why": "SURVEY 8d's 28 + 8*h bytes are the reference's walk; the canopy kernel climbs a 128 KiB LDS image "
                                  "and reads a's understory pre-summed, so those bytes never cross the fabric (results are bit-exact "
                                  "all the same: `parity`)" if info["strategy"] == "canopy" else
                                  "SURVEY 8d's 28 + 8*h bytes per pair over the kernel time"}
    roof.update({"algorithmic_bytes_per_pair": bytes_per_pair, "mean_path_edges": h_mean,
                 "algorithmic_GBps": achieved, "algorithmic_frac_of_hbm_peak": alg_frac,
                 "required_bytes_per_pair": required, "required_GBps": required * rate / 1e9,
                 "required_frac": required * rate / 1e9 / HBM_PEAK_GBPS})
    sweep = footprint_sweep()
    if sweep:
        roof["hbm_regime"] = sweep
    roof["note"] = ("bound hbm: achieved / peak / frac = fabric bytes the kernel moves per second (counters) against the 8 TB/s HBM3E peak. "
                    "At this tree's %.0f MiB gather footprint most record sectors are served by the 256 MiB Infinity Cache, which the "
                    "TCC_EA counters include; hbm_regime shows the same kernel family on trees whose records exceed it. "
                    "secondary_ceiling (added when the in-process microbenchmark ran): fabric read requests per second against the "
                    "random-64-byte-sector rate at the kernel's footprint -- what actually limits a gather kernel below the byte peak."
                    % (gather_footprint_bytes(info, 1 << args.levels) / 2**20))
    if info["strategy"] == "canopy":
        # SURVEY 8d asks for the achieved occupancy next to the fraction: the canopy kernels run one 1024-lane
        # workgroup per CU when the LDS image exceeds 80 KiB (two below that)
        image = ((info["canopy_nodes"] + 1) // 2) * 16
        wg = 1 if image > 80 * 1024 else 2
        roof["occupancy"] = {"waves_per_cu": 16 * wg, "max_waves_per_cu": 32, "lds_image_bytes": image,
                             "why": "%d workgroup(s) of 1024 lanes per CU beside a canopy image of %d KiB in LDS" % (wg, image // 1024)}
    line = {
        "metric": METRIC,
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "balanced 2^%d-leaf tree (%d nodes), %s, int64 ids in HBM -> float64 distance + "
                               "int32 MRCA id%s"
                               % (args.levels, n_nodes,
                                  ("%d uniform random leaf pairs per GPU per step" % n) if args.weak else
                                  ("one batch of %d uniform random leaf pairs per step, sharded over %d GPU(s)" % (n, world)),
                                  "" if (args.weak or world == 1) else
                                  {"root": " assembled on rank 0 inside the timed region",
                                   "allgather": " assembled on every rank inside the timed region",
                                   "none": "; every rank keeps its slice (nothing travels inside the timed region)"}[mode]),
                   "pairs_per_step": n_job, "pairs_per_gpu": pairs_this_rank if strong else n,
                   "tree_levels": args.levels, "kernel_family": info["strategy"],
                   "canopy_nodes": info["canopy_nodes"], "record_bytes": info["record_bytes"],
                   "gather": mode if (strong and world > 1) else None,
                   "sharding": ({"root": "contiguous pair slices (rank 0: %.0f %% of the batch, the peers share the rest: its kernels "
                                         "end when their transfers do), tree replicated, no data-path collective; results to rank 0 "
                                         "by RCCL send/recv over xGMI (%s on the wire, %d pieces per slice)",
                                 "allgather": "even contiguous pair slices (rank 0: %.0f %%), tree replicated, no data-path collective; every "
                                              "rank sends its result pieces to every other rank by grouped RCCL send/recv over xGMI (%s on "
                                              "the wire, %d pieces per slice)",
                                 "none": "even contiguous pair slices (rank 0: %.0f %%), tree replicated, nothing travels: float64 + int32 "
                                         "results stay on the GPU that computed them (%s would be the wire format; %d launches per slice)"}[mode]
                                % (100.0 * pairs_this_rank / max(n, 1),
                                   "float32 + 24-bit MRCA id = 7 bytes per pair" if wire_bytes == 7 else "float32 + int32 = 8 bytes per pair", plan.chunks))
                   if (strong and world > 1) else "none" if world == 1 else
                   "weak: every rank its own batch, tree replicated, nothing gathered"},
        "roofline": roof,
        "kernel_only_pairs_per_s": (n if strong else n * world) / (kernel_ms_max * 1e-3),
        "kernel_pairs_per_s_per_gpu": pairs_this_rank / (kernel_ms * 1e-3),
        "checksum": checksum,
    }
    if strong and world > 1:
        # what the gather costs on top of the slowest rank's kernels (exposed, after overlap)
        line["gather"] = mode
        line["gather_ms"] = max(0.0, ms_per_step - kernel_ms_max)
        line["wire_bytes_per_pair"] = wire_bytes
        line["gather_bytes_into_root"] = 0 if mode == "none" else wire_bytes * (n - pairs_this_rank)
        line["root_share"] = pairs_this_rank / max(n, 1)
        if calib:
            line["root_share_calibration"] = calib
    return line


def closing_wait(dg, rank, timeout_s, release=False):
    """While rank 0 runs the CPU baseline and the side legs (a minute or two) the peers wait on the
    process group's key-value store -- on the host, with a generous timeout -- instead of inside an
    RCCL barrier kernel that spins on their GPUs and is subject to the collective timeout.  (Without
    access to the store -- a torch that does not expose it -- the closing barrier itself is the wait.)"""
    import datetime
    try:
        store = dg.distributed_c10d._get_default_store()
    except Exception:      # noqa: BLE001 -- private API: fall back to the barrier in main()
        return
    if rank == 0:
        if release:
            store.set("suchtree_bench_rank0_done", "1")
    elif not release:
        store.wait(["suchtree_bench_rank0_done"], datetime.timedelta(seconds=max(60.0, timeout_s)))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    # Both start forms (self-launch parent -> ranks, and ranks started by the driver's torch.distributed.run)
    # run under the same HSA IPC mode: this pool's host driver only supports dmabuf IPC, and RCCL's
    # peer-to-peer setup (hipIpcGetMemHandle) fails under the legacy mode.  Set before torch / HIP load;
    # an explicit setting in the environment wins.  
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher's parent (before torch / HIP)
        sys.exit(self_launch(args, argv))

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
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL code path is used even with one rank,
    # so that it can be exercised on a 1-GPU box
    distributed = "RANK" in os.environ or world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("MASTER_PORT is not set: start the ranks with torch.distributed.run (or plain "
                                 "`python bench.py --gpus N`, which picks a free port)")
            os.environ["MASTER_PORT"] = str(_free_port())      # one rank: any free port will do
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        # a collective that involves every rank before the first batched point-to-point call
        # (torch.distributed.batch_isend_irecv: "if this is the first collective call in the
        # group ... all ranks must participate"): build the communicator here, deterministically
        dist_.barrier()

    from suchtree_amd import synth
    parent, dist = synth.balanced_tree(args.levels)
    be = HipBackend(args, parent, dist, local_rank)
    line = run_job(args, be, dist_ if distributed else None, world, rank, parent, dist,
                   peers_wait=(lambda: closing_wait(dist_, rank, args.launch_timeout)) if distributed else None)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if distributed:
        closing_wait(dist_, rank, args.launch_timeout, release=True)
        dist_.barrier()
        dist_.destroy_process_group()
    be.close()


if __name__ == "__main__":
    main()
