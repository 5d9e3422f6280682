"""Condense the rocprofv3 --pmc passes of scripts/sector_ceiling_counters.sh (gpurun_out/ceiling_cnt/) into
profiles/ceiling_counters_<tag>.json: per launch shape of the random-sector microbenchmark, what one lane read costs in L1-miss
requests, fabric requests, TA busy and TCP stall cycles, and the rates per second (durations from the same csv rows).
    python3 scripts/sector_ceiling_summary.py <out dir> <table MiB> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

out_dir, mib, tag = sys.argv[1], float(sys.argv[2]), sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    key, span = {}, {}
    for r in csv.DictReader(open(f)):
        if "k_gather" not in r["Kernel_Name"]:
            continue
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        key[d] = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        span[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for d, c in per.items():
        for k, v in c.items():
            acc[key[d]][k].append(v)
        dur[key[d]].append(span[d])
res = {"table_MiB": mib, "shapes": [],
       "what": "rocprofv3 --pmc passes of suchtree_amd/csrc/microbench.hip::k_gather (random 32-byte reads, one per 64-byte sector) at the headline "
               "kernel's gather footprint: what one lane read of the ceiling microbenchmark costs in L1-miss requests (TCP_TCC_READ_REQ) and fabric "
               "requests (TCC_EA0_RDREQ); durations are those of the launches under the counters (the first launch of a shape warms the caches and "
               "is included: medians)"}
for k, c in sorted(acc.items()):
    reads = k[1] * 256.0      # Grid_Size lanes x 256 reads each
    ns = sorted(dur[k])[len(dur[k]) // 2]
    row = {"kernel": k[0], "grid_lanes": k[1], "workgroup": k[2], "lane_reads": reads, "median_ns": ns, "Greads_per_s": reads / ns, "per_lane_read": {}}
    print(k, "lane reads %.3e  median %.1f us (under counters) = %.1f G reads/s" % (reads, ns / 1e3, reads / ns))
    for name, v in sorted(c.items()):
        m = sum(v) / len(v)
        row["per_lane_read"][name] = m / reads
        print("   %-34s %.4g  = %.3f per lane read" % (name, m, m / reads))
    res["shapes"].append(row)
if res["shapes"]:
    res["best"] = max(res["shapes"], key=lambda r: r["Greads_per_s"])
    json.dump(res, open(os.path.join(ROOT, "profiles", "ceiling_counters_%s.json" % tag), "w"), indent=1, sort_keys=True)
