"""Per-call latency of small requests through the facade (mailbox path vs staged pipe)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suchtree_amd import SuchTree

T = SuchTree(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "test.tree")).to_device()
dev = T._device_tree()
ids = T.leaf_node_ids
for mailbox in (1, 0):
    dev.set_option("small_batch_path", mailbox)
    for n in (1, 100, 1000, 2048):
        pairs = np.random.default_rng(1).choice(ids, size=(n, 2)).astype(np.int64)
        for _ in range(200):
            T.distances_bulk(pairs)
        t0 = time.perf_counter()
        reps = 2000
        for _ in range(reps):
            T.distances_bulk(pairs)
        dt = (time.perf_counter() - t0) / reps
        print("mailbox=%d n=%5d  %7.2f us per distances_bulk call" % (mailbox, n, dt * 1e6))
    t0 = time.perf_counter()
    for _ in range(2000):
        T.distance(0, 26)
    print("mailbox=%d scalar distance(): %7.2f us per call" % (mailbox, (time.perf_counter() - t0) / 2000 * 1e6))
    t0 = time.perf_counter()
    for _ in range(2000):
        T.common_ancestor("Ttal", "Oche")
    print("mailbox=%d scalar common_ancestor(): %7.2f us per call" % (mailbox, (time.perf_counter() - t0) / 2000 * 1e6))
