"""BASELINE config 4 across the GPUs of one node: the full lower-triangle distance matrix of
an m-leaf tree, sharded by equal pair counts (suchtree_amd.sharding.triangle_shard_bounds),
generated on each device and streamed to host memory in tiles.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        scripts/triangle_multi_gpu.py --leaves 100000 [--keep]

One process per GPU; no data-path collective (every rank owns a contiguous slice of the
4,999,950,000 pair indices); RCCL only for the barrier / max-time reduction / checksum.
Prints one JSON line from rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=int, default=100_000)
    ap.add_argument("--tile", type=int, default=1 << 26, help="pairs per host tile")
    ap.add_argument("--keep", action="store_true", help="keep the whole slice in host memory (needs RAM)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    distributed = "RANK" in os.environ
    if distributed:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from suchtree_amd import _capi, sharding, synth
    m = args.leaves
    parent, dist_ = synth.complete_tree(m, seed=44)
    tree = _capi.DeviceTree(parent, dist_, device=local)
    ids = np.arange(0, 2 * m, 2, dtype=np.int64)
    lo, hi = sharding.triangle_shard_bounds(m, world, rank)
    buf = np.empty(hi - lo if args.keep else min(args.tile, hi - lo), dtype=np.float64)
    tree.triangle_host(ids, k_begin=lo, k_count=min(args.tile, hi - lo), out_dist=buf[: min(args.tile, hi - lo)])  # warm-up
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    checksum = 0.0
    k = lo
    while k < hi:
        c = min(args.tile, hi - k)
        out = buf[k - lo: k - lo + c] if args.keep else buf[:c]
        tree.triangle_host(ids, k_begin=k, k_count=c, out_dist=out)
        checksum += float(out.sum())
        k += c
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed, checksum], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, checksum = float(tmax[0].item()), float(t[1].item())
    if rank == 0:
        total = m * (m - 1) // 2
        print(json.dumps({"workload": "full lower triangle of a %d-leaf tree streamed to host" % m,
                          "pairs": total, "n_gpus": world, "seconds": elapsed, "pairs_per_s": total / elapsed,
                          "host_GBps": total * 8 / elapsed / 1e9, "checksum": checksum,
                          "slice_of_rank0": [lo, hi]}), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    tree.close()


if __name__ == "__main__":
    main()
