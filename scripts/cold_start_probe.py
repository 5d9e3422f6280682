"""Cold start (GPU box): import, first tree on the GPU, first calls -- what a short script pays once."""
import os
import sys
import time

t0 = time.perf_counter()
import numpy as np   # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t1 = time.perf_counter()
from suchtree_amd import SuchTree, _capi   # noqa: E402
t2 = time.perf_counter()
T = SuchTree(os.path.join(ROOT, "tests", "golden", "gopher_louse", "host.tree") if os.path.exists(os.path.join(ROOT, "tests", "golden", "gopher_louse", "host.tree")) else os.path.join(ROOT, "tests", "golden", "host.tree"))
t3 = time.perf_counter()
ids = np.array(list(T.leaf_node_ids), dtype=np.int64)
pairs = np.stack([ids[:-1], ids[1:]], 1)
d = T.distances_bulk(pairs)
t4 = time.perf_counter()
d = T.distances_bulk(pairs)
t5 = time.perf_counter()
big = np.random.default_rng(1).choice(ids, size=(1_000_000, 2))
d = T.distances_bulk(big)
t6 = time.perf_counter()
d = T.distances_bulk(big)
t7 = time.perf_counter()
print("import numpy %.2f s | import suchtree_amd %.2f s | SuchTree(newick) %.3f s | first small call (GPU start, tables, kernels loaded) %.3f s | second %.1f us | "
      "first 1e6-pair call (staging pipe set up) %.3f s | second %.2f ms"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, (t5 - t4) * 1e6, t6 - t5, (t7 - t6) * 1e3))
z = np.load(os.path.join(ROOT, "tests", "golden", "ml_tree.npz"))
t8 = time.perf_counter()
M = _capi.DeviceTree(z["parent"], z["distance"])
t9 = time.perf_counter()
print("ml.tree handle (tables + upload + kernel timing) in a warm process: %.3f s" % (t9 - t8))
