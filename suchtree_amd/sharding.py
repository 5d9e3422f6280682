"""Multi-GPU use of the hot path: one process per GPU, pairs sharded, tree replicated.

Pairs are independent, so the path shards with no data-path collective
(SURVEY.md section 8e): rank g computes the contiguous slice
``[g*n/G, (g+1)*n/G)`` on its own GPU against its own copy of the tree.  The only
communication is the final gather of the result slices (RCCL all-gather over
xGMI when the process group is ``nccl``; ``gloo`` in the CPU tests).

The reference has no counterpart (its only parallel recipe is a fork pool over
chunks, docs/examples/SuchTree_examples.md:462-497); the slice boundaries here
are that recipe's contiguous chunks.
"""
from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice of ``n`` items owned by ``rank`` (sizes differ by at most 1)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return (n * rank) // world, (n * (rank + 1)) // world


def triangle_shard_bounds(m: int, world: int, rank: int) -> Tuple[int, int]:
    """Slice of the ``m(m-1)/2`` lower-triangle pair indices k = i(i-1)/2 + j owned by
    ``rank``: equal pair counts, i.e. row boundaries proportional to sqrt(g/G)."""
    return shard_bounds(m * (m - 1) // 2, world, rank)


def triangle_row_of(k):
    """Row i of the lower-triangle index k = i(i-1)/2 + j (0 <= j < i), vectorised, exact."""
    k = np.asarray(k, dtype=np.int64)
    i = ((1.0 + np.sqrt(1.0 + 8.0 * k.astype(np.float64))) / 2.0).astype(np.int64)
    # fix floating-point rounding at the row boundaries
    i = np.where(i * (i - 1) // 2 > k, i - 1, i)
    i = np.where((i + 1) * i // 2 <= k, i + 1, i)
    return i


def distances_sharded(tree, pairs, group=None, gather: bool = True,
                      compute: Optional[Callable] = None):
    """Distances and MRCA ids for ``pairs`` with the work split over the process group.

    Every rank passes the same ``pairs`` (host int64 (n,2)); each computes its
    slice on its own GPU; with ``gather=True`` every rank returns the full
    ``(dist float64[n], mrca int32[n])``, otherwise its own slice plus bounds.

    ``compute(pairs_slice) -> (dist, mrca)`` defaults to
    ``tree.distances_and_ancestors_bulk`` (the HIP path).  The CPU test-suite
    injects a checker here because without a GPU the product path refuses to run.
    """
    import torch
    import torch.distributed as dist

    if compute is None:
        compute = tree.distances_and_ancestors_bulk
    pairs = np.asarray(pairs)
    n = int(pairs.shape[0])
    if group is None and not dist.is_initialized():
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    if hi > lo:
        d, m = compute(pairs[lo:hi])
    else:
        d, m = np.zeros(0, dtype=np.float64), np.zeros(0, dtype=np.int32)
    if not gather:
        return d, m, (lo, hi)
    if world == 1:
        return d, m
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    width = (n + world - 1) // world          # slices differ by at most one: pad to a common width
    buf_d = torch.zeros(width, dtype=torch.float64, device=device)
    buf_m = torch.zeros(width, dtype=torch.int32, device=device)
    buf_d[: hi - lo] = torch.from_numpy(np.ascontiguousarray(d)).to(device)
    buf_m[: hi - lo] = torch.from_numpy(np.ascontiguousarray(m)).to(device)
    all_d = torch.empty(world * width, dtype=torch.float64, device=device)
    all_m = torch.empty(world * width, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(all_d, buf_d, group=group)
    dist.all_gather_into_tensor(all_m, buf_m, group=group)
    all_d = all_d.cpu().numpy().reshape(world, width)
    all_m = all_m.cpu().numpy().reshape(world, width)
    out_d = np.empty(n, dtype=np.float64)
    out_m = np.empty(n, dtype=np.int32)
    for g in range(world):
        glo, ghi = shard_bounds(n, world, g)
        out_d[glo:ghi] = all_d[g, : ghi - glo]
        out_m[glo:ghi] = all_m[g, : ghi - glo]
    return out_d, out_m


def distances_sharded_device(tree, pairs, group=None, gather: bool = True):
    """Device-resident form: ``pairs`` is an int64 (n,2) CUDA/HIP tensor present on every
    rank's GPU; each rank computes its contiguous slice with the HIP kernels and the result
    slices are all-gathered GPU-to-GPU (RCCL over xGMI) -- nothing touches the host.
    Returns ``(dist float64[n], mrca int32[n])`` tensors on this rank's GPU (or this rank's
    slice and its bounds with ``gather=False``)."""
    import torch
    import torch.distributed as dist
    from . import torch_interop

    n = int(pairs.shape[0])
    if group is None and not dist.is_initialized():
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    d, m = torch_interop.distances_device(tree, pairs[lo:hi])
    if not gather:
        return d, m, (lo, hi)
    if world == 1:
        return d, m
    width = (n + world - 1) // world
    buf_d = torch.zeros(width, dtype=torch.float64, device=pairs.device)
    buf_m = torch.zeros(width, dtype=torch.int32, device=pairs.device)
    buf_d[: hi - lo] = d
    buf_m[: hi - lo] = m
    all_d = torch.empty(world * width, dtype=torch.float64, device=pairs.device)
    all_m = torch.empty(world * width, dtype=torch.int32, device=pairs.device)
    dist.all_gather_into_tensor(all_d, buf_d, group=group)
    dist.all_gather_into_tensor(all_m, buf_m, group=group)
    if n == world * width:
        return all_d, all_m
    out_d = torch.empty(n, dtype=torch.float64, device=pairs.device)
    out_m = torch.empty(n, dtype=torch.int32, device=pairs.device)
    for g in range(world):
        glo, ghi = shard_bounds(n, world, g)
        out_d[glo:ghi] = all_d[g * width: g * width + (ghi - glo)]
        out_m[glo:ghi] = all_m[g * width: g * width + (ghi - glo)]
    return out_d, out_m
