"""Random 64-byte-sector read ceiling by table footprint (GPU box): what bounds the walk family on
trees whose lineage tables are GBs (the 1e6-leaf depth-338 tree gathers from 3.4 GB)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_legs   # noqa: E402

lib = bench_legs._micro()
g = ctypes.c_double(0)
for mib in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    best = None
    for unroll, blocks, threads in bench_legs.SECTOR_SHAPES:
        rc = lib.stmb_random_sector_reads_shape(0, mib << 20, 32, unroll, blocks, threads, 2, ctypes.byref(g))
        if rc != 0:
            print("rc", rc)
            break
        if best is None or g.value > best[0]:
            best = (g.value, unroll, blocks, threads)
    print("table %5d MiB  best %.1f G sector reads/s  (unroll %d, blocks %d, threads %d)" % ((mib,) + best), flush=True)
