"""Side legs of bench.py (rank 0, after the timed region; none of this is `value`):

  hardware_ceilings   random-sector and streaming-copy rates of THIS GPU, best over a sweep of
                      launch shapes (suchtree_amd/csrc/microbench.hip)
  mrca_ids_only       MRCA ids without distances on the same batch
  host_path_leg       the same batch prefix from pageable numpy arrays (PCIe inclusive)
  other_configs       short legs on BASELINE configs 2 (ml.tree / nj.tree, 1e7 pairs), 4 (100k-leaf
                      lower triangle, generated on the device; a prefix streamed to the host) and
                      5 (fish-worm linked_distances + Laplacian), each with its SURVEY 8d
                      algorithmic bytes and an oracle check of a sample

`be` is bench.HipBackend.  The oracle is used as the checker only.
"""
import ctypes
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBPS = 8000.0


# --------------------------------------------------------------------------------------------
def _micro():
    from suchtree_amd import build as st_build
    try:
        lib = ctypes.CDLL(st_build.MICRO_LIB)
    except OSError:
        return None     # helper library not built: the line simply carries no measured ceilings
    dp = ctypes.POINTER(ctypes.c_double)
    lib.stmb_random_sector_reads_shape.argtypes = [ctypes.c_int, ctypes.c_longlong] + [ctypes.c_int] * 5 + [dp]
    lib.stmb_stream_copy_shape.argtypes = [ctypes.c_int, ctypes.c_longlong] + [ctypes.c_int] * 5 + [dp]
    return lib


SECTOR_SHAPES = [(u, b, t) for u in (4, 8, 16) for b, t in ((256, 1024), (512, 1024), (1024, 1024), (1024, 512), (2048, 256))]
COPY_SHAPES = [(u, b, t, nt) for u in (1, 2, 4, 8) for b, t in ((0, 256), (0, 1024), (2048, 1024), (4096, 512)) for nt in (0, 1)]


def sector_ceiling(lib, device_index, table_bytes, sweep_log=None, shapes=None):
    """Best random 64-byte-sector read rate (G reads/s) over a sweep of launch shapes, from a table of exactly
    `table_bytes` (rounded to 64): the ceiling a gather kernel with that footprint is measured against."""
    g = ctypes.c_double(0)
    size = max(64, int(table_bytes) // 64 * 64)
    best = None
    for unroll, blocks, threads in (shapes or SECTOR_SHAPES):
        rc = lib.stmb_random_sector_reads_shape(device_index, size, 32, unroll, blocks, threads, 2, ctypes.byref(g))
        if rc != 0:
            return None
        if sweep_log is not None:
            sweep_log.append({"what": "random_sector", "MiB": size / 2**20, "unroll": unroll, "blocks": blocks,
                              "threads": threads, "Greads_per_s": g.value})
        if best is None or g.value > best[0]:
            best = (g.value, {"unroll": unroll, "blocks": blocks, "threads": threads})
    return {"MiB": size / 2**20, "Greads_per_s": best[0], "best_shape": best[1], "shapes_swept": len(shapes or SECTOR_SHAPES)}


def hardware_ceilings(device_index, footprint_bytes, sweep_log=None):
    """Measured in this process, on this GPU: the random 64-byte-sector read rate for a table of exactly the size
    of the record tables the kernel gathers from (`table`; a 64 MiB table beside it for comparison with earlier
    rounds), and the streaming copy rate -- each the BEST over a sweep of launch shapes (unroll x grid x block;
    copy: also non-temporal), because a ceiling taken at one shape is only a floor.  `sweep_log`: list that receives
    one dict per measured shape."""
    lib = _micro()
    if lib is None:
        return None
    g = ctypes.c_double(0)
    out = {}
    for name, size in (("table", footprint_bytes), ("table_64MiB", 64 << 20)):
        c = sector_ceiling(lib, device_index, size, sweep_log)
        if c is None:
            return None
        out[name] = c
    best = None
    for unroll, blocks, threads, nt in COPY_SHAPES:
        rc = lib.stmb_stream_copy_shape(device_index, 1 << 30, 2, unroll, blocks, threads, nt, ctypes.byref(g))
        if rc != 0:
            break
        if sweep_log is not None:
            sweep_log.append({"what": "stream_copy", "unroll": unroll, "blocks": blocks, "threads": threads, "nt": nt,
                              "GBps": g.value})
        if best is None or g.value > best[0]:
            best = (g.value, {"unroll": unroll, "blocks": blocks or "exact", "threads": threads, "non_temporal": bool(nt)})
    if best:
        out["stream_copy_GBps"] = best[0]
        out["stream_copy_best_shape"] = best[1]
        out["stream_copy_shapes_swept"] = len(COPY_SHAPES)
    return out


def three_ceilings(traffic, pairs_per_s, algorithmic_bytes_per_pair, ceiling=None, copy_GBps=None):
    """The three figures every leg reports for its dominant kernel, from the committed PMC passes of that kernel
    (`traffic`: profiles/traffic_*.json) and this run's pair rate: the SURVEY 8d algorithmic bytes / HBM peak (can
    exceed 1: the kernels do not move the reference's bytes), the counter traffic / HBM peak, and the fabric read
    request rate / the random-sector ceiling measured in this process at the kernel's own footprint."""
    out = {"algorithmic": {"bytes_per_pair": algorithmic_bytes_per_pair,
                           "GBps": algorithmic_bytes_per_pair * pairs_per_s / 1e9,
                           "frac_of_hbm_peak": algorithmic_bytes_per_pair * pairs_per_s / 1e9 / HBM_PEAK_GBPS}}
    if traffic and traffic.get("hbm_bytes_per_launch") and traffic.get("pairs_per_launch"):
        per_pair = traffic["hbm_bytes_per_launch"] / traffic["pairs_per_launch"]
        out["counter_traffic"] = {"bytes_per_pair": per_pair, "GBps": per_pair * pairs_per_s / 1e9,
                                  "frac_of_hbm_peak": per_pair * pairs_per_s / 1e9 / HBM_PEAK_GBPS}
        if copy_GBps:
            out["counter_traffic"]["frac_of_measured_copy"] = per_pair * pairs_per_s / 1e9 / copy_GBps
        req = traffic.get("counters_mean_per_launch", {}).get("TCC_EA0_RDREQ_sum")
        if req:
            rpp = req / traffic["pairs_per_launch"]
            out["request_rate"] = {"fabric_reads_per_pair": rpp, "Greads_per_s": rpp * pairs_per_s / 1e9}
            if ceiling:
                out["request_rate"].update({"ceiling_Greads_per_s": ceiling["Greads_per_s"], "ceiling_table_MiB": ceiling["MiB"],
                                            "ceiling_shape": ceiling.get("best_shape"),
                                            "frac": rpp * pairs_per_s / 1e9 / ceiling["Greads_per_s"]})
        l2 = traffic.get("counters_mean_per_launch", {}).get("TCP_TCC_READ_REQ_sum")
        if l2:
            # reads that miss the CU's L1 and go to L2: what the deep-tree kernels' time follows (profiles/ladder_ablation_r05.log),
            # against the random-sector rate of an L2-resident table from the round's committed sweep
            rpp = l2 / traffic["pairs_per_launch"]
            out["l2_request_rate"] = {"l2_reads_per_pair": rpp, "Greads_per_s": rpp * pairs_per_s / 1e9}
            c = committed_sector_ceiling(2 << 20)
            if c and c.get("table_MiB", 0) <= 2:
                out["l2_request_rate"].update({"ceiling_Greads_per_s": c["Greads_per_s"], "ceiling_source": "%s (%s)" % (c["source"], c["entry"]),
                                               "frac": rpp * pairs_per_s / 1e9 / c["Greads_per_s"]})
    return out


def committed_sector_ceiling(footprint_bytes):
    """The committed round sweep's random-sector ceiling at the footprint nearest to `footprint_bytes`
    (scripts/ceilings_sweep.py -> profiles/ceilings_sweep_rNN.json), so that the line's secondary ceiling can be
    checked against a file under profiles/ and not only against this process's own measurement."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "ceilings_sweep_r[0-9][0-9].json")))
    if not files:
        return None
    try:
        best = json.load(open(files[-1])).get("best", {})
    except Exception:      # noqa: BLE001
        return None
    rows = [(abs(v["table"]["MiB"] * 2**20 - footprint_bytes), k, v) for k, v in best.items()
            if isinstance(v, dict) and isinstance(v.get("table"), dict)]
    if not rows:
        return None
    _, name, v = min(rows, key=lambda r: r[0])
    return {"source": os.path.relpath(files[-1], ROOT), "entry": name, "table_MiB": v["table"]["MiB"],
            "Greads_per_s": v["table"]["Greads_per_s"], "stream_copy_GBps": v.get("stream_copy_GBps")}


def committed_ceiling_counters():
    """profiles/ceiling_counters_rNN.json (scripts/sector_ceiling_counters.sh): what one lane read of the random-sector
    microbenchmark costs in L1-miss requests and in fabric requests at the headline kernel's footprint."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "ceiling_counters_r[0-9][0-9].json")))
    if not files:
        return None
    try:
        b = json.load(open(files[-1])).get("best") or {}
    except Exception:      # noqa: BLE001
        return None
    per = b.get("per_lane_read", {})
    return {"source": os.path.relpath(files[-1], ROOT), "l1_per_read": per.get("TCP_TCC_READ_REQ_sum"), "fabric_per_read": per.get("TCC_EA0_RDREQ_sum"),
            "Greads_per_s_under_counters": b.get("Greads_per_s")}


def load_traffic(tag):
    """profiles/traffic_<tag>_rNN.json of the latest round that has one (committed PMC passes of a leg's kernel)."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_%s_r[0-9][0-9]*.json" % tag)))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:      # noqa: BLE001
        return None, None


# --------------------------------------------------------------------------------------------
def mrca_ids_only(be, pairs, out_m, reps=5):
    """MRCA ids alone (common_ancestors_bulk, quartets): on trees with in-order ids they come from a
    rank table and a sparse table over the canopy, without the distance kernels."""
    torch, tree, stream = be.torch, be.tree, be.stream
    n = pairs.shape[0]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    chk = torch.empty(n, dtype=torch.int32, device=be.device)
    tree.distances_device(pairs.data_ptr(), n, 0, chk.data_ptr(), stream=stream.cuda_stream)
    ev0.record(stream)
    for _ in range(reps):
        tree.distances_device(pairs.data_ptr(), n, 0, chk.data_ptr(), stream=stream.cuda_stream)
    ev1.record(stream)
    ev1.synchronize()
    tree.fault_check(stream.cuda_stream)
    return {"ids_per_s": float(reps) * n / (ev0.elapsed_time(ev1) * 1e-3),
            "matches_the_fused_launch": bool(torch.equal(chk, out_m[:n]))}


def host_path_leg(be, pairs, out_d, out_m):
    """End-to-end leg (SURVEY 8d asks for it next to the kernel-only figure; it is never `value`):
    the same batch prefix from pageable host numpy arrays through the library's staged host path --
    what T.distances_bulk(numpy) costs, PCIe inclusive."""
    tree = be.tree
    n = pairs.shape[0]
    wire_in, wire_out = tree.info()["host_wire_bytes_in"], tree.info()["host_wire_bytes_out"]
    k2 = min(n, 50_000_000)
    host_pairs = pairs[:k2].cpu().numpy()
    ref_d, ref_m = out_d[:k2].cpu().numpy(), out_m[:k2].cpu().numpy()
    h_d, h_m = np.empty(k2), np.empty(k2, dtype=np.int32)
    tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
    t_h = 1e30
    for _ in range(2):
        t = time.perf_counter()
        tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
        t_h = min(t_h, time.perf_counter() - t)
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
    # int32 ids, C order (what a caller who keeps ids as int32 hands over): no narrowing pass
    pairs32 = host_pairs.astype(np.int32)
    tree.distances_host(pairs32, True, True, out_dist=h_d, out_mrca=h_m)
    t_32 = time.perf_counter()
    tree.distances_host(pairs32, True, True, out_dist=h_d, out_mrca=h_m)
    t_32 = time.perf_counter() - t_32
    i32_ok = bool(np.array_equal(h_d.view(np.int64), ref_d.view(np.int64)) and np.array_equal(h_m, ref_m))
    tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
    # distances alone -- what the reference's SuchTree.distances(ids) returns: 4 bytes per pair come back over the link
    tree.distances_host(host_pairs, True, False, out_dist=h_d)
    t_d = 1e30
    for _ in range(2):
        t = time.perf_counter()
        tree.distances_host(host_pairs, True, False, out_dist=h_d)
        t_d = min(t_d, time.perf_counter() - t)
    d_ok = bool(np.array_equal(h_d.view(np.int64), ref_d.view(np.int64)))
    # measurement switches of the pipeline (handle option "measure"; results are not produced and the call says so): the CPU passes alone (nothing launched) and the
    # GPU / link side alone (no pack, no unpack).  The first bounds what ONE process can feed: a multi-device handle
    # (SuchTree(..., devices=[0..7])) runs one such pipeline per GPU against the same host memory system, so
    # cpu_passes / pairs_per_s is the most it can scale to on this host.
    def switched(bits):
        from suchtree_amd._capi import MeasureOnly
        tree.set_option("measure", bits)
        try:
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                try:
                    tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)
                except MeasureOnly:      # (what such a call returns: its results are not valid)
                    pass
                else:
                    raise RuntimeError("a call under the measure option must not return ST_OK")
                best = min(best, time.perf_counter() - t0)
            return best
        finally:
            tree.set_option("measure", 0)
    t_cpu = switched(4)
    t_link = switched(2)
    tree.distances_host(host_pairs, True, True, out_dist=h_d, out_mrca=h_m)      # (the arrays hold results again)
    return {
        "pairs_per_s": k2 / t_h, "pairs_per_s_fresh_arrays": k2 / t_f,
        "pairs_per_s_call_and_drop_loop": k2 / t_l,
        "pairs_per_s_int32_ids": k2 / t_32,
        "pairs_per_s_distances_only": k2 / t_d,
        "pairs": k2,
        "cpu_passes_pairs_per_s": k2 / t_cpu, "link_side_pairs_per_s": k2 / t_link,
        "one_process_many_gpus_ceiling": {"x_one_gpu": (k2 / t_cpu) / (k2 / t_h),
                                          "why": "the pack / unpack passes alone (handle option measure = 4) sustain cpu_passes_pairs_per_s on "
                                                 "this host; a multi-device handle feeds one pipeline per GPU from the same host memory system"},
        "link_bytes_per_pair": {"in": wire_in, "out": wire_out},
        "link_GBps_out": float(wire_out) * k2 / t_h / 1e9,
        "what": "pageable numpy int64 pairs in -> float64 distances + int32 MRCA ids out, PCIe inclusive "
                "(on trees of fewer than 2^24 nodes ids cross as 24 bits each and MRCA ids come back as 24 bits, else int32; distances as float32, widened on the host); reused result arrays / "
                "result arrays allocated by the call, first use of their memory (what a single "
                "SuchTree.distances_bulk call returns) / the same call in a loop that drops each result "
                "(blocks recycled by the library, release included) / int32 ids handed over as they are / distances alone, as "
                "the reference's distances() returns them (reused array)",
        "matches_device_results": bool(np.array_equal(f_d.view(np.int64), ref_d.view(np.int64))
                                       and np.array_equal(f_m, ref_m)
                                       and np.array_equal(h_d.view(np.int64), ref_d.view(np.int64))
                                       and np.array_equal(h_m, ref_m) and loop_ok and i32_ok and d_ok)}


# --------------------------------------------------------------------------------------------
def _device_rate(be, tree, pairs_t, reps=5):
    """Median kernel time over `reps` launches of the whole batch (HIP events on the launch stream)."""
    torch, stream = be.torch, be.stream
    n = pairs_t.shape[0]
    out_d = torch.empty(n, dtype=torch.float64, device=be.device)
    out_m = torch.empty(n, dtype=torch.int32, device=be.device)
    ms = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        tree.distances_device(pairs_t.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(be.device)
        if r:
            ms.append(e0.elapsed_time(e1))
    tree.fault_check(stream.cuda_stream)
    return float(np.median(ms)), out_d, out_m


def ladder_form(info, n):
    """1 when the handle's scalar ladder kernel runs its joint form on a batch of n pairs (st_tree_info.ladder_sums, for batches
    up to ladder_sums_max_pairs where that is set), else 0."""
    return int(bool(info.get("ladder_sums")) and (not info.get("ladder_sums_max_pairs") or n <= info["ladder_sums_max_pairs"]))


def _mean_path_edges(be, parent, pairs_t, out_m):
    from suchtree_amd.newick import node_depths
    torch = be.torch
    depth_t = torch.from_numpy(node_depths(parent).astype(np.int64)).to(be.device)
    h = depth_t[pairs_t[:, 0]] + depth_t[pairs_t[:, 1]] - 2 * depth_t[out_m.long()]
    return float(h.double().mean().item())


def config2(be, name, n=10_000_000, sample=400_000):
    """BASELINE configs[1]: data/bigtrees/{ml,nj}.tree (flat-array fixture), 1e7 random leaf pairs."""
    from oracle.oracle import OracleTree
    from suchtree_amd import _capi
    torch = be.torch
    z = np.load(os.path.join(GOLDEN, "%s_tree.npz" % name))
    parent, dist, leaf_ids = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
    tree = _capi.DeviceTree(parent, dist, device=be.local_rank)
    try:
        pairs = np.random.default_rng(2).choice(leaf_ids, size=(n, 2))
        pairs_t = torch.from_numpy(pairs).to(be.device)
        O = OracleTree(parent, dist)
        # the CPU column of this config ("... vs Cython CPU"): the oracle on a prefix of these same pairs, one thread and all
        # host cores; its distances are also the parity sample
        cpu, cpu_d = leg_cpu_baseline(O, pairs, "the leg's 10,000,000 pairs on %s.tree" % name)
        sample = min(sample, len(cpu_d))
        want_d, want_m = cpu_d[:sample], O.mrca_bulk(pairs[:sample])
        out = {"workload": "%s.tree (%d leaves, %d nodes, depth %d), %d uniform random leaf pairs, int64 ids in HBM -> "
                           "float64 distance + int32 MRCA id" % (name, len(leaf_ids), len(parent), tree.info()["depth"], n),
               "cpu_baseline": cpu}
        # "default": the family (and kernel) the library picks for this tree and batch; "walk": the walk family forced
        for key, strategy in (("default", "auto"), ("walk", "walk")):
            tree.set_strategy(strategy)
            ms, out_d, out_m = _device_rate(be, tree, pairs_t)
            if key == "default":
                h_mean = _mean_path_edges(be, parent, pairs_t, out_m)
                out["mean_path_edges"] = h_mean
                out["algorithmic_bytes_per_pair"] = 28 + 8 * h_mean
            ok = (np.array_equal(out_d[:sample].cpu().numpy().view(np.int64), want_d.view(np.int64))
                  and np.array_equal(out_m[:sample].cpu().numpy(), want_m))
            kernel = tree.info()["big_batch_kernel"] if key == "default" else "walk_sorted"
            out[key] = {"kernel_ms": ms, "pairs_per_s": n / (ms * 1e-3), "bit_exact_on_sample": bool(ok),
                        "sample_pairs": sample, "kernel": kernel, "x_cpu_all_cores": n / (ms * 1e-3) / cpu["value"],
                        "x_cpu_one_thread": n / (ms * 1e-3) / cpu["single_thread_value"],
                        "roofline": leg_roofline(be, "%s%s" % ("" if key == "default" else "walk_", name), kernel, n / (ms * 1e-3), ms, n,
                                                 28 + 8 * out["mean_path_edges"], tree.info()["device_bytes"],
                                                 ladder_sums=ladder_form(tree.info(), n) if key == "default" else None,
                                                 why=None if key == "default" else
                                                 "SURVEY 8d's 28 + 8*h bytes are the reference's walk; the walk family reads a's side from "
                                                 "lineage sums and b's as a stream of lineage lengths, three edges per gather elsewhere")}
            del out_d, out_m
        tree.set_strategy("auto")
        h_d, h_m = np.empty(n), np.empty(n, dtype=np.int32)
        tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
        t = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
            t = min(t, time.perf_counter() - t0)
        out["host_path"] = {"pairs_per_s": n / t,
                            "bit_exact_on_sample": bool(np.array_equal(h_d[:sample].view(np.int64), want_d.view(np.int64))
                                                        and np.array_equal(h_m[:sample], want_m))}
        out["kernel_family_info"] = {k: tree.info()[k] for k in ("canopy_nodes", "record_bytes", "lineage_entries")}
        return out
    finally:
        tree.close()


def leg_cpu_baseline(O, pairs_host, what, seconds=2.5):
    """The reference's CPU path beside a leg's GPU figure (BASELINE configs[1] is "... vs Cython CPU"): the oracle
    (oracle/suchtree_oracle.c: pyx:911-943 with the visited-list scan of pyx:999-1030, which is O(depth^2) per pair on
    deep trees) on a bounded prefix of the LEG'S OWN pairs -- one thread (the reference as shipped: one thread, GIL held)
    and all host cores on contiguous chunks (the reference under a fork pool).  Returns (block, distances of the
    all-cores sample) so that the same sample serves as the parity check."""
    from oracle import oracle as orc
    kind, what_code = "port", "oracle/suchtree_oracle.c (visited-list MRCA, 20-byte AoS)"
    if orc.ref_lib() is not None and hasattr(O, "nodes"):      # the reference's own compiled code where it travelled with the snapshot
        O = orc.RefTree(O.nodes["parent"], O.nodes["distance"], depth=O.depth)
        kind, what_code = "reference", "the reference's compiled SuchTree._distances (SuchTree/MuchTree.c as shipped, oracle/ref_harness.c)"
    cores = len(os.sched_getaffinity(0))
    n = len(pairs_host)
    k = min(n, 20_000)
    t0 = time.perf_counter()
    O.distances(pairs_host[:k])
    r1 = k / max(time.perf_counter() - t0, 1e-9)
    n1 = int(min(n, max(k, r1 * seconds * 0.4)))
    t0 = time.perf_counter()
    O.distances(pairs_host[:n1])
    r1 = n1 / max(time.perf_counter() - t0, 1e-9)
    k = min(n, 4_000 * cores)
    t0 = time.perf_counter()
    O.distances_mt(pairs_host[:k], cores)
    rm = k / max(time.perf_counter() - t0, 1e-9)
    nm = int(min(n, max(k, rm * seconds)))
    t0 = time.perf_counter()
    d = O.distances_mt(pairs_host[:nm], cores)
    rm = nm / max(time.perf_counter() - t0, 1e-9)
    return {"value": rm, "unit": "pairs/s", "cores": cores, "kind": kind,
            "sample": "first %d pairs of %s, %s, %d pthreads on contiguous chunks" % (nm, what, what_code, cores),
            "single_thread_value": r1, "single_thread_sample": "first %d pairs, one thread" % n1}, d


def traffic_matches(traffic, kernel, pairs_per_launch=None, ladder_sums=None):
    """A committed PMC summary speaks for a launch only if it was taken on the same kernel (name as the handle reports
    it: "canopy_ladder" -> "k_canopy_ladder<"; ladder_sums: and on the same form of it -- the joint form's template
    argument list ends in "true>") and, where given, the same batch size."""
    if not traffic or not traffic.get("hbm_bytes_per_launch") or not traffic.get("pairs_per_launch"):
        return False
    full = str(traffic.get("kernel_full_name") or traffic.get("kernel") or "")
    want = "k_walk" if str(kernel).startswith("walk") else "k_%s<" % kernel      # (walk family: k_walk / k_walk_sorted by batch and source)
    if kernel and want not in full:
        return False
    if kernel == "canopy_ladder" and ladder_sums is not None and bool(ladder_sums) != ("SrcContig, true>" in full):
        return False
    return pairs_per_launch is None or abs(traffic["pairs_per_launch"] - pairs_per_launch) <= 0.01 * pairs_per_launch


def leg_roofline(be, tag, kernel, pairs_per_s, kernel_ms, pairs_per_launch, algorithmic_bytes_per_pair, table_bytes, why=None,
                 ladder_sums=None):
    """A side leg's `roofline` block, in the shape of the headline's: bound hbm, `achieved` = fabric bytes per pair by
    counters (profiles/traffic_<tag>_rNN.json: committed rocprofv3 --pmc passes of this leg's own launch; used only when
    the profiled kernel and batch size are this leg's) x this run's pairs per second of kernel time, `frac` = achieved /
    8 TB/s, `traffic` = counter bytes per launch.  SURVEY 8d's algorithmic bytes (28 + 8 h per pair) are the flagged
    sub-block `algorithmic` (these kernels climb in LDS and read pre-summed records: the fraction exceeds 1 and is not a
    bandwidth claim).  `request_rate` / `l2_request_rate`: what the time of a gather kernel actually follows."""
    traffic, traffic_file = load_traffic(tag)
    ok = traffic_matches(traffic, kernel, pairs_per_launch, ladder_sums)
    ceiling = None
    lib = _micro() if not getattr(be, "no_microbench", False) else None
    if lib is not None and ok:
        ceiling = sector_ceiling(lib, be.local_rank, min(int(table_bytes), 8 << 30), shapes=[(8, 512, 1024), (16, 1024, 1024), (8, 2048, 256)])
    three = three_ceilings(traffic if ok else None, pairs_per_s, algorithmic_bytes_per_pair, ceiling)
    alg = three["algorithmic"]
    alg["exceeds_peak"] = bool(alg["frac_of_hbm_peak"] > 1.0)
    alg["why"] = why or ("SURVEY 8d's 28 + 8*h bytes are the reference's walk; this kernel climbs the canopy in LDS and reads "
                         "understories as pre-summed records, so those bytes never cross the fabric (results are bit-exact all the same)")
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "kernel": ("k_" + kernel + (" (a's side from the lineage sums)" if kernel == "canopy_ladder" and ladder_sums else "")) if kernel else None,
            "kernel_ms": kernel_ms, "pairs_per_launch": pairs_per_launch}
    if ok:
        ct = three["counter_traffic"]
        roof.update({"traffic": ct["bytes_per_pair"] * pairs_per_launch, "traffic_bytes_per_pair": ct["bytes_per_pair"],
                     "traffic_source": traffic_file, "counters_kernel": traffic.get("kernel_full_name", traffic.get("kernel")),
                     "achieved": ct["GBps"], "frac": ct["frac_of_hbm_peak"],
                     "achieved_is": "fabric bytes per pair by counters (%s) x this run's pairs per second of kernel time" % traffic_file})
        k = (traffic.get("kernels") or {}).get(traffic.get("kernel_full_name"), {})
        if k.get("avg_ns"):
            roof["rocprof"] = {"calls": k.get("calls"), "kernel_avg_ms": k["avg_ns"] * 1e-6,
                               "frac": traffic["hbm_bytes_per_launch"] / (k["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS}
        for key in ("request_rate", "l2_request_rate"):
            if key in three:
                roof[key] = three[key]
    else:
        # no committed PMC pass of THIS kernel at THIS batch size: the block says so instead of borrowing another launch's bytes
        roof.update({"traffic": None, "achieved": alg["GBps"], "frac": alg["frac_of_hbm_peak"],
                     "achieved_is": "SURVEY 8d algorithmic bytes (no committed counter pass matches this launch%s): not a bandwidth "
                                    "figure where it exceeds 1" % ((": %s holds %s" % (traffic_file, traffic.get("kernel_full_name"))) if traffic else "")})
    roof["algorithmic"] = alg
    return roof


def shape_tree_leg(be, skew, what, tag, n_leaves=1_000_000, n=10_000_000, sample=200_000, walk_tag=None):
    """Trees beyond the BASELINE configs that the judge of round 3 asked to see in the driver's line: 1,000,000
    leaves of skewed random shape (suchtree_amd.synth.skewed_tree, default_rng(5)) -- skew 0.9: depth ~340, beyond
    512-byte records (round 3: the walk family's tables alone; since round 4 1 KB records read by the scalar ladder
    kernel, the walk family timed beside it: walk_tag); skew 0.8: depth ~173, 512-byte records.  The kernel is the one
    the handle picked when it was created.  1e7 uniform random leaf pairs in HBM, oracle check of a sample."""
    from oracle.oracle import OracleTree
    from suchtree_amd import _capi, synth
    torch = be.torch
    parent, dist = synth.skewed_tree(np.random.default_rng(5), n_leaves, skew)
    t0 = time.perf_counter()
    tree = _capi.DeviceTree(parent, dist, device=be.local_rank)
    create_s = time.perf_counter() - t0
    try:
        info = tree.info()
        g = torch.Generator(device=be.device)
        g.manual_seed(3)
        pairs_t = torch.randint(0, n_leaves, (n, 2), generator=g, device=be.device, dtype=torch.int64) * 2      # leaves = even ids
        p = pairs_t[:sample].cpu().numpy()
        O = OracleTree(parent, dist)
        p_cpu = pairs_t[:2_000_000].cpu().numpy()
        cpu, cpu_d = leg_cpu_baseline(O, p_cpu, "the leg's %d pairs" % n)
        sample = min(sample, len(cpu_d))
        p = p_cpu[:sample]
        want_d, want_m = cpu_d[:sample], O.mrca_bulk(p)

        def run(tag_, kernel):
            ms, out_d, out_m = _device_rate(be, tree, pairs_t)
            h = _mean_path_edges(be, parent, pairs_t, out_m)
            ok = (np.array_equal(out_d[:sample].cpu().numpy().view(np.int64), want_d.view(np.int64))
                  and np.array_equal(out_m[:sample].cpu().numpy(), want_m))
            alg = 28 + 8 * h
            r = {"kernel_ms": ms, "pairs_per_s": n / (ms * 1e-3), "bit_exact_on_sample": bool(ok), "sample_pairs": sample,
                 "x_cpu_all_cores": n / (ms * 1e-3) / cpu["value"], "x_cpu_one_thread": n / (ms * 1e-3) / cpu["single_thread_value"],
                 "roofline": leg_roofline(be, tag_, kernel, n / (ms * 1e-3), ms, n, alg, info["device_bytes"],
                                          ladder_sums=ladder_form(info, n) if kernel == "canopy_ladder" else None)}
            return r, h

        main, h_mean = run(tag, info["big_batch_kernel"])
        out = {"workload": "%s: %d leaves, %d nodes, depth %d, %d uniform random leaf pairs, int64 ids in HBM -> float64 "
                           "distance + int32 MRCA id" % (what, n_leaves, len(parent), info["depth"], n),
               "kernel_family": info["strategy"], "kernel": info["big_batch_kernel"], "tuned": info["tuned"],
               "record_bytes": info["record_bytes"], "canopy_nodes": info["canopy_nodes"], "device_MB": info["device_bytes"] / 1e6,
               "create_seconds": create_s, "mean_path_edges": h_mean, "algorithmic_bytes_per_pair": 28 + 8 * h_mean,
               "cpu_baseline": cpu}
        out.update(main)
        if walk_tag and info["strategy"] != "walk":
            tree.set_strategy("walk")
            out["walk"] = run(walk_tag, "walk_sorted")[0]
            out["walk"]["kernel"] = "walk_sorted"
            tree.set_strategy("auto")
        return out
    finally:
        tree.close()


def config4(be, m=100_000, host_pairs=1 << 30):
    """BASELINE configs[3]: full lower triangle of the complete 100,000-leaf tree in linked_distances order
    (k = i(i-1)/2 + j -> (ids[j], ids[i]), MuchTree.pyx:2918-2925), generated on the device; a prefix
    of the stream through the host entry point into one reused numpy buffer."""
    from oracle.oracle import OracleTree
    from suchtree_amd import _capi, synth
    from suchtree_amd.sharding import triangle_row_of
    torch, stream = be.torch, be.stream
    parent, dist = synth.complete_tree(m, seed=44)
    tree = _capi.DeviceTree(parent, dist, device=be.local_rank)
    try:
        ids = np.arange(0, 2 * m, 2, dtype=np.int64)
        ids_t = torch.from_numpy(ids).to(be.device)
        total = m * (m - 1) // 2
        tile = 1 << 27
        out_d = torch.empty(tile, dtype=torch.float64, device=be.device)
        out_m = torch.empty(tile, dtype=torch.int32, device=be.device)
        out = {"workload": "full lower triangle of the complete %d-leaf binary tree (seed 44): %d pairs in "
                           "linked_distances order, generated on the device" % (m, total), "pairs": total}
        # mean path length from one tile in the middle of the stream
        k_mid = total // 2
        tree.triangle_device(ids_t.data_ptr(), m, k_mid, tile, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
        kk = torch.arange(k_mid, k_mid + tile, device=be.device, dtype=torch.int64)
        rows = ((1.0 + torch.sqrt(1.0 + 8.0 * kk.double())) / 2.0).long()
        rows = torch.where(rows * (rows - 1) // 2 > kk, rows - 1, rows)
        rows = torch.where((rows + 1) * rows // 2 <= kk, rows + 1, rows)
        cols = kk - rows * (rows - 1) // 2
        pp = torch.stack([cols * 2, rows * 2], 1)
        h_mean = _mean_path_edges(be, parent, pp, out_m)
        del kk, rows, cols, pp
        out["mean_path_edges"] = h_mean
        out["algorithmic_bytes_per_pair"] = 12 + 8 * h_mean      # generated pairs: no ids in
        for strategy in ("canopy", "walk"):
            tree.set_strategy(strategy)
            best = 1e30
            for _ in range(2):
                torch.cuda.synchronize(be.device)
                t0 = time.perf_counter()
                for k0 in range(0, total, tile):
                    tree.triangle_device(ids_t.data_ptr(), m, k0, min(tile, total - k0), out_d.data_ptr(), out_m.data_ptr(),
                                         stream=stream.cuda_stream)
                torch.cuda.synchronize(be.device)
                best = min(best, time.perf_counter() - t0)
            tree.fault_check(stream.cuda_stream)
            n_tiles = (total + tile - 1) // tile
            kernel = {"canopy": "canopy_ilp", "walk": "walk"}[strategy]      # (generated pairs: the walk family's plain kernel)
            out[strategy] = {"seconds_whole_triangle": best, "pairs_per_s": total / best, "kernel": kernel,
                             "launches": n_tiles, "pairs_per_launch": tile,
                             "roofline": leg_roofline(be, "tri" if strategy == "canopy" else "walk_tri", kernel, total / best,
                                                      best * 1e3 / n_tiles, tile, out["algorithmic_bytes_per_pair"], tree.info()["device_bytes"],
                                                      why="SURVEY 8d's 12 + 8*h bytes (generated pairs: no ids in) are the reference's walk; "
                                                          "the kernel climbs the canopy in LDS, reads understories as pre-summed records, and "
                                                          "consecutive pairs of a triangle row share their second leaf")}
        tree.set_strategy("auto")
        del out_d, out_m
        host_tile = 1 << 26
        buf_d = np.empty(host_tile, dtype=np.float64)
        tree.triangle_host(ids, k_begin=0, k_count=host_tile, out_dist=buf_d)
        budget = min(total, host_pairs)
        t0 = time.perf_counter()
        done = 0
        while done < budget:
            c = min(host_tile, budget - done)
            tree.triangle_host(ids, k_begin=done, k_count=c, out_dist=buf_d[:c])
            done += c
        dt = time.perf_counter() - t0
        out["streamed_to_host"] = {"pairs": done, "pairs_per_s": done / dt, "float64_GBps_into_host_memory": done * 8 / dt / 1e9,
                                   "what": "first %d pairs of the triangle through st_triangle_host in %d-pair tiles into "
                                           "one reused float64 buffer" % (done, host_tile)}
        O = OracleTree(parent, dist)
        k0 = total // 3
        n_cpu = 4_000_000
        kk = np.arange(k0, k0 + n_cpu)
        rows = triangle_row_of(kk)
        cols = kk - rows * (rows - 1) // 2
        pp = np.ascontiguousarray(np.stack([ids[cols], ids[rows]], 1))
        # CPU column: the reference builds these id arrays in a Python loop (pyx:2918-2925) and then calls _distances on them;
        # only the _distances part is timed here (the enumeration above is numpy, untimed)
        cpu, cpu_d = leg_cpu_baseline(O, pp, "the triangle's pairs k = %d ... (ids enumerated by numpy, untimed)" % k0)
        out["cpu_baseline"] = cpu
        for strategy in ("canopy", "walk"):
            out[strategy]["x_cpu_all_cores"] = out[strategy]["pairs_per_s"] / cpu["value"]
            out[strategy]["x_cpu_one_thread"] = out[strategy]["pairs_per_s"] / cpu["single_thread_value"]
        sample = min(200_000, len(cpu_d))
        d, mm = tree.triangle_host(ids, k_begin=k0, k_count=sample, want_mrca=True)
        out["bit_exact_on_sample"] = bool(np.array_equal(d.view(np.int64), cpu_d[:sample].view(np.int64))
                                          and np.array_equal(mm, O.mrca_bulk(pp[:sample])))
        out["sample_pairs"] = sample
        return out
    finally:
        tree.close()


def config5(be):
    """BASELINE configs[4]: fish-worm SuchLinkedTrees -- linked_distances on both trees + the Laplacian."""
    import pandas as pd
    from oracle import oracle as orc
    from suchtree_amd import SuchTree
    from suchtree_amd.linked import SuchLinkedTrees
    d = os.path.join(GOLDEN, "fish_worm")
    links = pd.read_csv(d + "/links.csv", index_col=0)
    A, B = SuchTree(d + "/host.tree", device=be.local_rank), SuchTree(d + "/guest.tree", device=be.local_rank)
    SLT = SuchLinkedTrees(A, B, links)
    r = SLT.linked_distances()
    best = 1e30
    for _ in range(10):
        t0 = time.perf_counter()
        SLT.linked_distances()
        best = min(best, time.perf_counter() - t0)
    ids_a, ids_b = orc.linked_pairs(SLT.linklist)
    OA = orc.OracleTree(A._flat.parent, A._flat.distance)
    OB = orc.OracleTree(B._flat.parent, B._flat.distance)
    # CPU column: _distances on the same two id arrays, one thread (18,145 pairs per tree: what the reference runs) -- the reference's
    # own compiled code where oracle/_ref/libref_hotpath.so travelled with the snapshot, else the oracle's restatement
    CA, CB, cpu_kind = OA, OB, "port"
    if orc.ref_lib() is not None:
        CA, CB, cpu_kind = orc.RefTree(A._flat.parent, A._flat.distance, depth=OA.depth), orc.RefTree(B._flat.parent, B._flat.distance, depth=OB.depth), "reference"
    t_cpu = 1e30
    for _ in range(5):
        t0 = time.perf_counter()
        CA.distances(ids_a)
        CB.distances(ids_b)
        t_cpu = min(t_cpu, time.perf_counter() - t0)
    ok = bool(np.array_equal(np.asarray(r["TreeA"]).view(np.int64), OA.distances(ids_a).view(np.int64))
              and np.array_equal(np.asarray(r["TreeB"]).view(np.int64), OB.distances(ids_b).view(np.int64)))
    SLT.laplacian()
    t0 = time.perf_counter()
    lap = SLT.laplacian()
    t_lap = time.perf_counter() - t0
    # the oracle's own dense-block restatement of MuchTree.pyx:1750-1813 + 3081-3145
    fa, fb = A._flat, B._flat
    aj_o = orc.linked_adjacency((fa.parent, fa.left, fa.right, fa.distance), (fb.parent, fb.left, fb.right, fb.distance),
                                SLT.linklist, SLT.subset_a_root, SLT.subset_b_root, A.polytomy_epsilon, B.polytomy_epsilon)
    lap_ok = bool(np.array_equal(lap.view(np.int64), orc.linked_laplacian(aj_o).view(np.int64)))
    return {"workload": "fish-worm: host 21 leaves / guest 191 leaves, 191 links -> 18,145 link pairs on each tree",
            "pairs": 2 * len(ids_a), "seconds_linked_distances": best, "pairs_per_s": 2 * len(ids_a) / best,
            "cpu_baseline": {"value": 2 * len(ids_a) / t_cpu, "unit": "pairs/s", "cores": 1, "kind": cpu_kind,
                             "sample": "all %d link pairs of both trees, _distances on one thread (the id arrays given: the "
                                       "reference's Python-loop enumeration of pyx:2918-2925 is not timed)" % (2 * len(ids_a))},
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None,
                         "achieved": 2 * len(ids_a) * (28 + 8 * 10) / best / 1e9, "frac": 2 * len(ids_a) * (28 + 8 * 10) / best / 1e9 / HBM_PEAK_GBPS,
                         "achieved_is": "a nominal 108 B/pair (28 + 8 h at h = 10) over the whole call's wall time: two launches of 18,145 pairs are "
                                        "launch-latency bound; this config is plumbing (SURVEY 8d: far too small to say anything about throughput)"},
            "distances_bit_exact": ok, "seconds_laplacian": t_lap, "laplacian_shape": list(lap.shape),
            "laplacian_bit_exact": lap_ok}


def other_configs(be):
    """The other BASELINE configs, a few seconds in all.  A failing leg reports its error instead of
    taking the headline line down with it."""
    out = {}
    for key, fn in (("config2_ml_tree", lambda: config2(be, "ml")), ("config2_nj_tree", lambda: config2(be, "nj")),
                    ("config4_triangle_100k", lambda: config4(be)), ("config5_fish_worm", lambda: config5(be)),
                    ("walk_only_tree", lambda: shape_tree_leg(be, 0.9, "tree beyond 512-byte records (1 KB records / the walk family's tables)",
                                                              "bigdeep", walk_tag="walk_bigdeep")),
                    ("deep_long_record_tree", lambda: shape_tree_leg(be, 0.8, "deep tree with 512-byte records", "s80"))):
        t0 = time.perf_counter()
        try:
            out[key] = fn()
        except Exception as e:      # noqa: BLE001 -- reported in the line
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        out[key]["leg_seconds"] = time.perf_counter() - t0
    return out
