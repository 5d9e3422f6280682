"""Host-thread time by phase of the host path (SUCHTREE_AMD_TRACE_PIPE), by batch size (GPU box)."""
import os
import sys
import time

import numpy as np

os.environ["SUCHTREE_AMD_TRACE_PIPE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suchtree_amd import SuchTree, synth   # noqa: E402

T = SuchTree(synth.balanced_tree(17))
T.to_device()
leaves = np.asarray(T.leaf_node_ids, dtype=np.int64)
rng = np.random.default_rng(1)
for n in [int(x) for x in (sys.argv[1:] or (30_000, 100_000, 300_000, 1_000_000, 3_000_000))]:
    pairs = rng.choice(leaves, size=(n, 2))
    for _ in range(6):
        t0 = time.perf_counter()
        T.distances_bulk(pairs)
        dt = time.perf_counter() - t0
    print("n=%d last call %.1f us" % (n, dt * 1e6), file=sys.stderr, flush=True)
