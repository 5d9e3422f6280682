"""The random-sector microbenchmark at one footprint and its best launch shapes, for rocprofv3 (round 6): which requests does the
ceiling of bench.py's secondary_ceiling consist of?  Run under `rocprofv3 --pmc ... --kernel-trace` (scripts/sector_ceiling_counters.sh).
    python3 scripts/sector_ceiling_probe.py <table MiB>"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import build as st_build      # noqa: E402

lib = ctypes.CDLL(st_build.MICRO_LIB)
dp = ctypes.POINTER(ctypes.c_double)
lib.stmb_random_sector_reads_shape.argtypes = [ctypes.c_int, ctypes.c_longlong] + [ctypes.c_int] * 5 + [dp]
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
size = int(mib * 2**20) // 64 * 64
g = ctypes.c_double(0)
for unroll, blocks, threads in ((8, 512, 1024), (16, 1024, 1024), (8, 2048, 256), (16, 256, 1024)):
    rc = lib.stmb_random_sector_reads_shape(0, size, 32, unroll, blocks, threads, 3, ctypes.byref(g))
    print("table %.0f MiB unroll %d blocks %d threads %d: rc %d  %.2f G reads/s  (%d lane reads per launch)" %
          (mib, unroll, blocks, threads, rc, g.value, blocks * threads * 256), flush=True)
