"""relative_evolutionary_divergence through the bulk path (GPU box): seconds and pairs per tree.
  python scripts/red_probe.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from suchtree_amd import SuchTree, synth
for name in ("ml", "nj", "bal16", "bal18"):
    if name in ("ml", "nj"):
        z = np.load("tests/golden/%s_tree.npz" % name); p, d = z["parent"], z["distance"]
    else:
        p, d = synth.balanced_tree(int(name[3:]))
    T = SuchTree((p, d))
    T.distance(0, 2)      # tables on the device
    calls = []
    bulk = T.distances_bulk
    def counted(pairs, _bulk=bulk, _calls=calls):
        t0 = time.perf_counter(); r = _bulk(pairs); _calls.append((len(pairs), time.perf_counter() - t0)); return r
    T.distances_bulk = counted
    t0 = time.perf_counter()
    red = T.relative_evolutionary_divergence
    t = time.perf_counter() - t0
    print("%-6s %8d nodes  %10d pairs in one batch (%.3f s on the path)  RED of every node in %.2f s" % (name, T.size, calls[0][0], calls[0][1], t), flush=True)
