"""The round's hardware-ceiling sweep (GPU box): every launch shape of bench_legs.hardware_ceilings -- random 64-byte-sector
reads and streaming copy -- at the gather footprints of the footprint sweep's three trees -> profiles/ceilings_sweep_<round>.json.
usage: python scripts/ceilings_sweep.py r05"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_legs   # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = {"best": {}, "sweep": []}
# 20 MiB: 2^20 leaves x (16 B of cherry records + 4 B rec_a4), 36 MiB without the cherry records; 272 / 1088 MiB: 2^22 / 2^24 leaves x (64 B + 4 B) without them
# 1 / 2 / 7 MiB: tables that stay in an XCD's 4 MiB L2 (the sparse table of nj.tree, the b records of ml.tree / nj.tree): the rate
# at which a CU's L1 misses are served from L2 -- what the deep-tree kernels run against (profiles/ladder_ablation_r05.log)
for name, size in (("headline_20MiB", 20 << 20), ("headline_without_cherries_36MiB", 36 << 20), ("leaves_2_22_272MiB", 272 << 20), ("leaves_2_24_1088MiB", 1088 << 20),
                   ("l2_resident_1MiB", 1 << 20), ("l2_resident_2MiB", 2 << 20), ("l2_partly_7MiB", 7 << 20)):
    log = []
    hw = bench_legs.hardware_ceilings(0, size, log)
    if hw is None:
        raise SystemExit("libst_microbench.so missing or failed")
    out["best"][name] = hw
    for e in log:
        e["footprint"] = name
    out["sweep"] += log
    print(name, "random sector %.2f G/s, copy %.0f GB/s" % (hw["table"]["Greads_per_s"], hw.get("stream_copy_GBps", 0)), flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "ceilings_sweep_%s.json" % tag), "w"), indent=1)
os.makedirs(os.path.join(ROOT, "gpurun_out", "round_profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "round_profiles", "ceilings_sweep_%s.json" % tag), "w"), indent=1)
