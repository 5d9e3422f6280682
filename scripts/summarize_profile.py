"""Condense a scripts/profile_gpu.sh output directory into the small files that
get committed under profiles/ (kernel stats table + HBM traffic per launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def kernel_stats(trace_dir):
    rows = []
    for f in find(trace_dir, "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel_durations(trace_dir):
    dur = defaultdict(list)
    for f in find(trace_dir, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return dur


def dominant_kernel(dur, substr):
    """full name of the kernel containing substr with the largest total duration"""
    best, best_t = None, -1
    for k, v in dur.items():
        if substr in k and sum(v) > best_t:
            best, best_t = k, sum(v)
    return best


def counter_means(pmc_dir, kernel_substr):
    """mean counter value per dispatch of kernels whose name contains kernel_substr"""
    acc = defaultdict(list)
    for f in find(pmc_dir, "*counter_collection.csv"):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            if kernel_substr not in r["Kernel_Name"]:
                continue
            per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for d in per_dispatch.values():
            for k, v in d.items():
                acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "k_canopy"
    pairs_per_launch = float(sys.argv[4]) if len(sys.argv) > 4 else 1e8
    # bytes of coalesced input stream per pair: 16 (int64 ids), 0 for pairs generated on the device (triangle)
    stream_bytes_per_pair = float(os.environ.get("STREAM_BYTES_PER_PAIR", "16"))
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(prof, exist_ok=True)
    lines = []
    stats = kernel_stats(os.path.join(out_dir, "trace"))
    if stats:
        cols = list(stats[0].keys())
        lines.append(",".join(cols))
        for r in stats:
            lines.append(",".join(str(r[c]) for c in cols))
    dur = per_kernel_durations(os.path.join(out_dir, "trace"))
    summary = {"tag": tag, "kernel": kernel, "kernels": {}, "pairs_per_launch": pairs_per_launch,
               "stream_bytes_per_pair": stream_bytes_per_pair}
    try:      # bench.py's own line in the trace pass's log says what was launched
        line = [l for l in open(os.path.join(out_dir, "trace.log")).read().splitlines() if l.lstrip().startswith('{"metric"')][-1]
        cfg = json.loads(line)["config"]
        summary["config"] = {"levels": cfg.get("tree_levels"), "canopy_nodes": cfg.get("canopy_nodes"), "record_bytes": cfg.get("record_bytes"),
                             "pairs_per_step": cfg.get("pairs_per_step"), "kernel_family": cfg.get("kernel_family")}
    except Exception:      # noqa: BLE001 -- not a bench.py run (scripts/tune_gpu.py legs)
        pass
    if os.environ.get("PROFILE_CONFIG"):      # what was launched (bench.py uses the counters only for a run of the same configuration)
        summary["config"] = json.loads(os.environ["PROFILE_CONFIG"])
    for k, v in dur.items():
        v2 = sorted(v)
        summary["kernels"][k] = {"calls": len(v), "avg_ns": sum(v) / len(v), "min_ns": v2[0], "max_ns": v2[-1]}
    full = dominant_kernel(dur, kernel)
    if full:
        summary["kernel_full_name"] = full
    counters = {}
    for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
        if os.path.isdir(d):
            means, counts = counter_means(d, full or kernel)
            counters.update(means)
    summary["counters_mean_per_launch"] = counters
    summary["counters_per_pair"] = {k: v / pairs_per_launch for k, v in counters.items()}
    # calibration: the same counters on launches of exactly known traffic
    # (scripts/micro/calib_requests.hip): bytes per read request of a coalesced 16-B-per-lane
    # stream and of random 32-byte reads that touch one 64-byte sector each
    cal = {}
    for d in sorted(glob.glob(os.path.join(out_dir, "cal_*"))):
        if os.path.isdir(d):
            for kern, key in (("k_copy", "copy"), ("k_gather", "gather")):
                means, _ = counter_means(d, kern)
                for k, v in means.items():
                    cal["%s.%s" % (key, k)] = v
    COPY_BYTES, GATHER_READS = float(1 << 30), 512.0 * 1024 * 256
    if cal.get("copy.TCC_EA0_RDREQ_sum") and cal.get("gather.TCC_EA0_RDREQ_sum"):
        cal["stream_bytes_per_read_request"] = COPY_BYTES / cal["copy.TCC_EA0_RDREQ_sum"]
        cal["gather_read_requests_per_read"] = cal["gather.TCC_EA0_RDREQ_sum"] / GATHER_READS
    if cal.get("copy.FETCH_SIZE"):
        cal["copy_bytes_per_FETCH_SIZE_KiB"] = COPY_BYTES / cal["copy.FETCH_SIZE"]
    if cal.get("gather.FETCH_SIZE") and cal.get("gather.TCC_EA0_RDREQ_sum"):
        cal["gather_FETCH_SIZE_bytes_per_request"] = cal["gather.FETCH_SIZE"] * 1024.0 / cal["gather.TCC_EA0_RDREQ_sum"]
    if cal.get("copy.WRITE_SIZE"):
        cal["copy_bytes_per_WRITE_SIZE_KiB"] = COPY_BYTES / cal["copy.WRITE_SIZE"]
    if cal:
        with open(os.path.join(prof, "calibration_%s.json" % tag), "w") as fh:
            json.dump(cal, fh, indent=1, sort_keys=True)
    summary["calibration"] = {k: v for k, v in cal.items() if not k.startswith(("copy.", "gather."))}
    pairs = summary["pairs_per_launch"]
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        # rocprofv3 reports both in KiB.  The guide's gfx950 rule (FETCH_SIZE x2) holds for wide
        # coalesced streams only: FETCH_SIZE is TCC_EA0_RDREQ x 64 B, a streamed request carries
        # 128 B, a single-sector record request 64 B.  Kept for reference:
        fetch = counters["FETCH_SIZE"] * 1024.0
        write = counters["WRITE_SIZE"] * 1024.0
        summary["hbm_bytes_per_launch_x2_rule"] = 2.0 * fetch + write
        summary["fetch_bytes_raw"] = fetch
        summary["write_bytes"] = write
    if "TCC_EA0_RDREQ_sum" in counters and "WRITE_SIZE" in counters:
        # fabric bytes by request size: the pair stream (16 B per pair, coalesced) leaves L2 as
        # requests of the calibrated stream size (128 B on gfx950); every other read request is
        # a single 64-byte sector of a record table; writes are coalesced streams (WRITE_SIZE exact)
        stream_req_bytes = cal.get("stream_bytes_per_read_request", 128.0)
        stream_requests = stream_bytes_per_pair * pairs / stream_req_bytes
        record_requests = max(0.0, counters["TCC_EA0_RDREQ_sum"] - stream_requests)
        summary["hbm_bytes_per_launch"] = stream_requests * stream_req_bytes + record_requests * 64.0 + counters["WRITE_SIZE"] * 1024.0
        summary["traffic_model"] = {"stream_read_requests": stream_requests, "stream_bytes_per_request": stream_req_bytes,
                                    "record_read_requests": record_requests, "record_bytes_per_request": 64.0,
                                    "write_bytes": counters["WRITE_SIZE"] * 1024.0,
                                    "record_requests_per_pair": record_requests / pairs}
    with open(os.path.join(prof, "kernel_stats_%s.csv" % tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(prof, "traffic_%s.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
