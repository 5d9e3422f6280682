"""profiles/traffic_fp<levels>_<round>.json (scripts/footprint_sweep.sh) -> profiles/footprint_sweep_<round>.json:
per tree the kernel's rocprofv3 average duration, the counter bytes per launch (request counts x calibrated bytes
per request + WRITE_SIZE, scripts/summarize_profile.py), both as a rate and as a fraction of the 8 TB/s HBM peak, next
to the bytes the pairs gather from (leaf records: rec_b half-records + the a side) and whether that set fits the
256 MiB Infinity Cache."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12
MALL = 256 << 20


def main():
    tag = sys.argv[1]
    levels = [int(a) for a in sys.argv[2:]] or [20, 22, 24]
    rows = []
    for L in levels:
        f = os.path.join(ROOT, "profiles", "traffic_fp%d_%s.json" % (L, tag))
        if not os.path.exists(f):
            continue
        t = json.load(open(f))
        full = t.get("kernel_full_name")
        k = t["kernels"].get(full, {})
        pairs = t["pairs_per_launch"]
        c = t.get("counters_mean_per_launch", {})
        info = {}
        log = os.path.join(ROOT, "gpurun_out", "prof_fp%d_%s" % (L, tag), "trace.log")
        if os.path.exists(log):      # tune_gpu.py prints tree.info() first
            for line in open(log):
                if line.startswith("{") and "record_bytes" in line:
                    info = eval(line, {"__builtins__": {}}, {})      # (a dict literal printed by our own script)
                    break
        n_leaves = 1 << L
        rec_half = info.get("b_table_bytes_per_leaf") or (info.get("record_bytes") or 0) // 2      # (bytes per leaf of the b-side table)
        a_bytes = info.get("a_side_bytes") or 8
        foot = n_leaves * (rec_half + a_bytes) if rec_half else None
        ns = k.get("avg_ns")
        b = t.get("hbm_bytes_per_launch")
        row = {"leaves": n_leaves, "levels": L, "kernel": full, "calls": k.get("calls"), "avg_ns": ns,
               "pairs_per_launch": pairs, "pairs_per_s": pairs / (ns * 1e-9) if ns else None,
               "record_bytes": info.get("record_bytes"), "a_side_bytes": a_bytes, "canopy_nodes": info.get("canopy_nodes"),
               "device_table_bytes": info.get("device_bytes"),
               "gather_footprint_bytes": foot, "fits_infinity_cache": (foot <= MALL) if foot else None,
               "counter_bytes_per_launch": b, "counter_bytes_per_pair": b / pairs if b else None,
               "counter_TBps": b / (ns * 1e-9) / 1e12 if (b and ns) else None,
               "frac_of_hbm_peak": b / (ns * 1e-9) / HBM_PEAK if (b and ns) else None,
               "fabric_read_requests_per_pair": c.get("TCC_EA0_RDREQ_sum", 0) / pairs if c.get("TCC_EA0_RDREQ_sum") else None,
               "read_requests_dram_per_pair": c.get("TCC_EA0_RDREQ_DRAM_sum", 0) / pairs if c.get("TCC_EA0_RDREQ_DRAM_sum") else None,
               "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if c.get("TCC_HIT_sum") and c.get("TCC_MISS_sum") else None,
               "useful_bytes_per_pair": 28 + (info.get("record_bytes") or 0) // 2 + (4 if a_bytes == 4 else 8) if rec_half else None,
               "source": "profiles/traffic_fp%d_%s.json, profiles/kernel_stats_fp%d_%s.csv" % (L, tag, L, tag)}
        rows.append(row)
    out = {"what": "st_distances_device on balanced trees, 1e8 uniform random leaf pairs (int64 ids in HBM -> float64 + int32), "
                   "SUCHTREE_AMD_AUTOTUNE=0; rocprofv3 --kernel-trace --stats and one --pmc pass per counter set",
           "hbm_peak_TBps": 8.0, "infinity_cache_bytes": MALL,
           "reading": "TCC_EA0 request counters sit at L2's memory side and include Infinity-Cache hits: below 256 MiB of gather "
                      "footprint the counter rate is fabric traffic, most of it served by the Infinity Cache; beyond it the same "
                      "counters are HBM traffic (plus the 28 B/pair streams, which are always HBM).",
           "trees": rows}
    path = os.path.join(ROOT, "profiles", "footprint_sweep_%s.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    for r in rows:
        print("2^%d leaves: %.3f ms, %.3e pairs/s, footprint %s MiB, %.1f B/pair by counters, %.2f TB/s = %.3f of peak, L2 hit %s"
              % (r["levels"], (r["avg_ns"] or 0) / 1e6, r["pairs_per_s"] or 0,
                 "%.0f" % (r["gather_footprint_bytes"] / 2**20) if r["gather_footprint_bytes"] else "?",
                 r["counter_bytes_per_pair"] or 0, r["counter_TBps"] or 0, r["frac_of_hbm_peak"] or 0, r["l2_hit_rate"]))


if __name__ == "__main__":
    main()
