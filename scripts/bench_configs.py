"""Throughput of the other BASELINE configs (bench.py covers configs[2], the headline).

    python scripts/bench_configs.py [--configs 1,2,4,5] [--out gpurun_out/configs.jsonl]

Prints one JSON line per measurement.  Config 4 (full lower triangle of a 100k-leaf
tree, 4,999,950,000 pairs) is generated on the device in tiles; "device" rows keep the
results in HBM (tile buffer overwritten), "host" rows stream every tile to host memory.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit(fh, **kw):
    line = json.dumps(kw)
    print(line, flush=True)
    if fh:
        fh.write(line + "\n")
        fh.flush()


def timed(fn, reps=3):
    fn()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,4,5")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tri-leaves", type=int, default=100_000)
    args = ap.parse_args()
    todo = set(args.configs.split(","))
    fh = open(args.out, "a") if args.out else None

    import torch
    from oracle.oracle import OracleTree
    from suchtree_amd import SuchTree, _capi, synth
    from suchtree_amd.linked import SuchLinkedTrees
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    cores = len(os.sched_getaffinity(0))
    G = os.path.join(ROOT, "tests", "golden")

    def device_rate(tree, pairs_t, reps=5):
        n = pairs_t.shape[0]
        out_d = torch.empty(n, dtype=torch.float64, device=dev)
        out_m = torch.empty(n, dtype=torch.int32, device=dev)
        ms = []
        for r in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            tree.distances_device(pairs_t.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                ms.append(e0.elapsed_time(e1))
        tree.fault_check(stream.cuda_stream)
        return n / (np.median(ms) * 1e-3), out_d, out_m

    if "1" in todo:
        T = SuchTree(os.path.join(G, "test.tree")).to_device()
        pairs = np.random.default_rng(1).choice(T.leaf_node_ids, size=(1000, 2)).astype(np.int64)
        O = OracleTree(T._flat.parent, T._flat.distance)
        assert np.array_equal(T.distances_bulk(pairs), O.distances(pairs))
        t = timed(lambda: T.distances_bulk(pairs), reps=20)
        t_cpu = timed(lambda: O.distances(pairs), reps=20)
        emit(fh, config=1, workload="gopher tree (29 nodes), 1000 random leaf pairs, numpy in/out through the facade",
             seconds_per_call=t, pairs_per_s=1000 / t, cpu_oracle_pairs_per_s=1000 / t_cpu, parity="bit-exact")

    if "2" in todo:
        for name in ("ml", "nj"):
            z = np.load(os.path.join(G, "%s_tree.npz" % name))
            parent, dist, leaf_ids = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
            tree = _capi.DeviceTree(parent, dist)
            n = 10_000_000
            pairs = np.random.default_rng(2).choice(leaf_ids, size=(n, 2))
            pairs_t = torch.from_numpy(pairs).to(dev)
            O = OracleTree(parent, dist)
            for strategy in ("canopy", "walk"):
                tree.set_strategy(strategy)
                rate, out_d, out_m = device_rate(tree, pairs_t)
                k = 400_000
                ok = (np.array_equal(out_d[:k].cpu().numpy().view(np.int64), O.distances(pairs[:k]).view(np.int64))
                      and np.array_equal(out_m[:k].cpu().numpy(), O.mrca_bulk(pairs[:k])))
                emit(fh, config=2, tree=name + ".tree", nodes=len(parent), pairs=n, kernel_family=strategy,
                     where="device-resident", pairs_per_s=rate, parity="bit-exact on %d" % k if ok else "MISMATCH",
                     info=tree.info())
            tree.set_strategy("canopy")
            h_d, h_m = np.empty(n), np.empty(n, dtype=np.int32)
            t = timed(lambda: tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m), reps=3)
            t_new = timed(lambda: tree.distances_host(pairs, True, True), reps=3)
            emit(fh, config=2, tree=name + ".tree", pairs=n, where="host numpy in/out (PCIe inclusive)",
                 pairs_per_s_reused_outputs=n / t, pairs_per_s_fresh_outputs_incl_alloc_and_free=n / t_new)
            s = 2_000_000
            t1 = timed(lambda: O.distances(pairs[: s // 8]), reps=1)
            tm = timed(lambda: O.distances_mt(pairs[:s], cores), reps=1)
            emit(fh, config=2, tree=name + ".tree", where="cpu oracle", single_thread_pairs_per_s=(s // 8) / t1,
                 all_cores_pairs_per_s=s / tm, cores=cores)
            tree.close()

    if "4" in todo:
        m = args.tri_leaves
        parent, dist = synth.complete_tree(m, seed=44)
        tree = _capi.DeviceTree(parent, dist)
        ids = np.arange(0, 2 * m, 2, dtype=np.int64)
        ids_t = torch.from_numpy(ids).to(dev)
        total = m * (m - 1) // 2
        tile = 1 << 27
        out_d = torch.empty(tile, dtype=torch.float64, device=dev)
        out_m = torch.empty(tile, dtype=torch.int32, device=dev)
        for strategy in ("canopy", "walk"):
            tree.set_strategy(strategy)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            acc = 0.0
            for k0 in range(0, total, tile):
                c = min(tile, total - k0)
                tree.triangle_device(ids_t.data_ptr(), m, k0, c, out_d.data_ptr(), out_m.data_ptr(),
                                     stream=stream.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tree.fault_check(stream.cuda_stream)
            emit(fh, config=4, workload="full lower triangle, complete %d-leaf binary tree (seed 44), generated on device" % m,
                 pairs=total, kernel_family=strategy, where="device (tile buffer overwritten)",
                 seconds=dt, pairs_per_s=total / dt, info=tree.info())
        tree.set_strategy("canopy")
        # streamed to host: the library's host entry point, tile by tile into one reused host buffer
        host_tile = 1 << 26
        buf_d = np.empty(host_tile, dtype=np.float64)
        t0 = time.perf_counter()
        done = 0
        budget_pairs = min(total, 1 << 30)      # bounded sample of the stream
        while done < budget_pairs:
            c = min(host_tile, budget_pairs - done)
            tree.triangle_host(ids, k_begin=done, k_count=c, out_dist=buf_d[:c])
            done += c
        dt = time.perf_counter() - t0
        emit(fh, config=4, where="streamed to host numpy (first %d pairs of the triangle)" % done,
             pairs=done, seconds=dt, pairs_per_s=done / dt, output_GBps=done * 8 / dt / 1e9)
        # parity on a slice
        O = OracleTree(parent, dist)
        k0 = total // 3
        d, mm = tree.triangle_host(ids, k_begin=k0, k_count=200_000, want_mrca=True)
        from suchtree_amd.sharding import triangle_row_of
        kk = np.arange(k0, k0 + 200_000)
        rows = triangle_row_of(kk)
        cols = kk - rows * (rows - 1) // 2
        pp = np.stack([ids[cols], ids[rows]], 1)
        ok = np.array_equal(d.view(np.int64), O.distances(pp).view(np.int64)) and np.array_equal(mm, O.mrca_bulk(pp))
        emit(fh, config=4, where="parity slice", parity="bit-exact on 200000" if ok else "MISMATCH")
        tree.close()

    if "host" in todo:
        # PCIe-inclusive rate of the headline workload: pageable numpy in, reused numpy out
        parent, dist = synth.balanced_tree(20)
        tree = _capi.DeviceTree(parent, dist)
        n = 100_000_000
        pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
        out_d = np.empty(n)
        out_m = np.empty(n, dtype=np.int32)
        t = timed(lambda: tree.distances_host(pairs, True, True, out_dist=out_d, out_mrca=out_m), reps=3)
        t_d = timed(lambda: tree.distances_host(pairs, True, False, out_dist=out_d), reps=3)
        t_new = timed(lambda: tree.distances_host(pairs, True, True), reps=2)
        emit(fh, config="3-host", workload="balanced 2^20-leaf tree, 1e8 pairs, host numpy in/out (PCIe inclusive)",
             pairs_per_s_reused_outputs=n / t, pairs_per_s_dist_only=n / t_d, pairs_per_s_fresh_outputs=n / t_new,
             GBps_over_pcie=n * 16 / t / 1e9, pcie_bytes_per_pair="8 in (int32 ids) + 4 (float32) + 4 (int32) out")
        tree.close()

    if "quartets" in todo:
        # row f4: quartet topologies (6 MRCAs per quartet), host numpy in/out
        for name in ("ml",):
            z = np.load(os.path.join(G, "%s_tree.npz" % name))
            parent, dist, leaf_ids = z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)
            T = SuchTree((parent, dist)).to_device()
            O = OracleTree(parent, dist)
            q = np.random.default_rng(8).choice(leaf_ids, size=(4_000_000, 4))
            got = T.quartet_topologies_bulk(q)
            ok = np.array_equal(got[:200_000], O.quartets(q[:200_000]))
            t = timed(lambda: T.quartet_topologies_bulk(q), reps=3)
            t_cpu = timed(lambda: O.quartets(q[:100_000]), reps=1)
            emit(fh, config="f4-quartets", tree=name + ".tree", quartets=len(q), quartets_per_s=len(q) / t,
                 cpu_oracle_quartets_per_s_single_thread=100_000 / t_cpu,
                 parity="bit-exact on 200000" if ok else "MISMATCH")

    if "5" in todo:
        import pandas as pd
        d = os.path.join(G, "fish_worm")
        links = pd.read_csv(d + "/links.csv", index_col=0)
        SLT = SuchLinkedTrees(SuchTree(d + "/host.tree"), SuchTree(d + "/guest.tree"), links)
        SLT.linked_distances()
        t = timed(lambda: SLT.linked_distances(), reps=10)
        t_lap = timed(lambda: SLT.laplacian(), reps=3)
        t_lap_np = timed(lambda: SLT.laplacian(on_gpu=False), reps=3)
        emit(fh, config=5, workload="fish-worm: 191 links -> 18145 link pairs on both trees", seconds_linked_distances=t,
             pairs_per_s=2 * 18145 / t, seconds_laplacian_422x422_gpu_assembly=t_lap,
             seconds_laplacian_422x422_numpy_assembly=t_lap_np)


if __name__ == "__main__":
    main()
