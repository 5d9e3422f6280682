"""Trees beyond the headline's 2^20 leaves (GPU box): balanced 2^22 / 2^24 leaves and a random-shape tree of
1.6e7 leaves -- creation time, device bytes, throughput of 1e8 random leaf pairs, parity of a sample."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleTree   # noqa: E402
from suchtree_amd import _capi, synth   # noqa: E402


def run(name, parent, dist, n=100_000_000):
    t0 = time.perf_counter()
    tree = _capi.DeviceTree(parent, dist)
    info = tree.info()
    print("%s: %d nodes, create %.1f s, %s, device tables %.2f GB" % (name, len(parent), time.perf_counter() - t0, info["strategy"], info["device_bytes"] / 1e9), flush=True)
    leaves = np.flatnonzero(np.bincount(parent[parent >= 0], minlength=len(parent)) == 0)
    g = torch.Generator(device="cuda").manual_seed(1)
    li = torch.from_numpy(leaves.astype(np.int64)).cuda()
    pairs = li[torch.randint(0, len(leaves), (n, 2), generator=g, device="cuda")]
    out_d = torch.empty(n, dtype=torch.float64, device="cuda")
    out_m = torch.empty(n, dtype=torch.int32, device="cuda")
    times = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr())
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    tree.fault_check()
    k = 20000
    O = OracleTree(parent, dist)
    ph = pairs[:k].cpu().numpy()
    ok = (np.array_equal(out_d[:k].cpu().numpy().view(np.int64), O.distances(ph).view(np.int64))
          and np.array_equal(out_m[:k].cpu().numpy(), O.mrca_bulk(ph)))
    print("    %d pairs: best %.2f ms = %.3e pairs/s, parity of %d pairs %s" % (n, min(times), n / min(times) * 1e3, k, "ok" if ok else "MISMATCH"), flush=True)
    # the host path on a slice
    m = 20_000_000
    hp = pairs[:m].cpu().numpy()
    d, mm = tree.distances_host(hp, True, True)
    t0 = time.perf_counter()
    d, mm = tree.distances_host(hp, True, True)
    dt = time.perf_counter() - t0
    okh = np.array_equal(d.view(np.int64), out_d[:m].cpu().numpy().view(np.int64)) and np.array_equal(mm, out_m[:m].cpu().numpy())
    print("    host path, %d pairs: %.3e pairs/s, equal to the device results: %s" % (m, m / dt, okh), flush=True)
    tree.close()
    del pairs, out_d, out_m
    torch.cuda.empty_cache()


for levels in [int(a) for a in sys.argv[1:]] or (22, 24):
    p, d = synth.balanced_tree(levels)
    run("balanced 2^%d" % levels, p, d)
    del p, d
p, d = synth.random_binary_tree(1 << 24, seed=5)
run("random shape, 2^24 leaves", p, d)
