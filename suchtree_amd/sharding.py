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


class ShardPlan:
    """Who computes what, and in which pieces the results travel.

    ``n`` pairs are cut into ``world`` contiguous slices, one per rank in rank order; every
    slice is cut again into ``chunks`` contiguous pieces so that the transfer of piece c to the
    root overlaps the kernel of piece c+1.

    ``root_share`` (default: 1/world, i.e. equal slices as in ``shard_bounds``): fraction of the
    pairs the root computes itself.  Results are assembled on the root over point-to-point
    links, and a result costs several times more to ship (8 B over one xGMI link) than to
    compute, so the fastest split gives the root MORE than 1/world: with kernel time t_k per
    pair and wire time t_w per pair and link, root and peers finish together at
    ``root_share = t_w / (t_w + (world - 1) * t_k)`` (``balanced_root_share``).
    """

    def __init__(self, n: int, world: int, rank: int, chunks: int = 4, root: int = 0,
                 root_share: Optional[float] = None, align: int = 1):
        if n < 0 or chunks < 1 or not (0 <= root < world) or align < 1:
            raise ValueError("bad plan")
        self.n, self.world, self.rank, self.chunks, self.root = int(n), int(world), int(rank), int(chunks), int(root)
        # piece boundaries inside a slice fall on multiples of `align` pairs from the slice's start (the packed
        # 24-bit id stream of a piece starts on a 4-byte boundary of the slice's buffer: align = 4)
        self.align = int(align)
        shard_bounds(n, world, rank)   # validates world / rank
        if root_share is None or world == 1:
            self.root_pairs = None
        else:
            if not (0.0 < root_share <= 1.0):
                raise ValueError("root_share must be in (0, 1]")
            self.root_pairs = min(self.n, max(0, int(round(root_share * self.n))))
        self.root_share = root_share

    def bounds(self, g: int) -> Tuple[int, int]:
        if self.root_pairs is None:
            return shard_bounds(self.n, self.world, g)
        if not (0 <= g < self.world):
            raise ValueError("bad rank")
        rest = self.n - self.root_pairs                 # split evenly over the world - 1 peers

        def size(r):
            if r == self.root:
                return self.root_pairs
            k = r if r < self.root else r - 1           # index among the peers
            lo, hi = shard_bounds(rest, self.world - 1, k)
            return hi - lo

        lo = sum(size(r) for r in range(g))
        return lo, lo + size(g)

    def piece(self, g: int, c: int) -> Tuple[int, int]:
        """Global pair range of piece ``c`` of rank ``g``'s slice (may be empty)."""
        lo, hi = self.bounds(g)
        plo, phi = shard_bounds(hi - lo, self.chunks, c)
        if self.align > 1:
            plo = plo // self.align * self.align
            phi = hi - lo if c == self.chunks - 1 else phi // self.align * self.align
        return lo + plo, lo + phi

    def pieces(self, g: int):
        return [self.piece(g, c) for c in range(self.chunks)]


WIRE_BYTES_PACKED = 7      # float32 distance + 24-bit MRCA id (trees of fewer than 2^24 nodes)
WIRE_BYTES_PLAIN = 8       # float32 distance + int32 MRCA id


def packed_bytes(n_pairs: int) -> int:
    """Bytes of ``n_pairs`` packed 24-bit ids, rounded up to whole dwords (the kernels store dwords)."""
    return (3 * int(n_pairs) + 3) // 4 * 4


def unpack_mrca24(packed, out):
    """``out[i]`` (int32) <- the 24-bit id at bytes [3 i, 3 i + 3) of the uint8 tensor ``packed``, 0xFFFFFF -> -1.
    Plain torch ops: works on CPU (gloo tests) and device tensors; the GPU bench passes the library's kernel
    (``st_unpack_mrca24_device``) to ``run_sharded`` instead."""
    import torch
    n = out.numel()
    b = packed[: 3 * n].view(n, 3).to(torch.int32)
    v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    out.copy_(torch.where(v == 0xFFFFFF, torch.full_like(v, -1), v))


def pack_mrca24(ids, packed):
    """The inverse (test backends and reference for the kernels' packing): int32 ids -> 3 bytes each, -1 -> 0xFFFFFF."""
    import torch
    n = ids.numel()
    v = ids.to(torch.int32) & 0xFFFFFF
    packed[: 3 * n].view(n, 3).copy_(torch.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], 1).to(torch.uint8))


def balanced_root_share(world: int, kernel_pairs_per_s: float, link_bytes_per_s: float,
                        wire_bytes_per_pair: float = 8.0) -> float:
    """Root's fraction of the batch for which its own kernels and every peer's transfers end
    together: the root computes s*n pairs in s*n*t_k; each peer ships (1-s)*n/(world-1) results
    over its own link in (1-s)*n/(world-1)*t_w (its kernels hide under the transfers)."""
    if world <= 1:
        return 1.0
    t_k = 1.0 / kernel_pairs_per_s
    t_w = wire_bytes_per_pair / link_bytes_per_s
    return t_w / (t_w + (world - 1) * t_k)


def run_sharded(plan: ShardPlan, compute: Callable, result_d, result_m, wire_d=None, wire_m=None,
                group=None, unpack: Optional[Callable] = None):
    """One pass of the hot path over ``plan.n`` pairs, sharded over the process group, with
    the result assembled on ``plan.root``: the north star's "pair batches shard across the
    GPUs, RCCL over xGMI only for the final gather".  The same code runs on ``nccl`` (RCCL,
    device tensors, asynchronous on streams) and on ``gloo`` (CPU tensors, the test-suite).

    ``compute(lo, hi, dst_d, dst_m)`` performs (or enqueues on the current stream) the pair
    computation for the global pair range ``[lo, hi)``: distances into the 1-D tensor
    ``dst_d`` (float64 or float32 -- the values are float32 sums either way), MRCA ids into
    ``dst_m``: an int32 tensor of length ``hi - lo``, or -- the packed wire format -- a uint8
    tensor of ``packed_bytes(hi - lo)`` bytes that receives them as 24 bits each.

    Wire format: ``wire_m`` of dtype uint8 (``sharded_buffers(..., packed_ids=True)``; trees of
    fewer than 2^24 nodes) = float32 + 24-bit id, **7 bytes per pair**, packed by the peers'
    kernels themselves (``st_distances_device_wire``), unpacked on the root piece by piece
    (``unpack(packed_piece, result_m_piece)``, default ``unpack_mrca24``); int32 = 8 bytes per
    pair, ids received straight into ``result_m``.  The plan must have ``align`` 4 in packed mode.

    Root:   ``result_d`` float64[n] and ``result_m`` int32[n] receive everything: its own
            slice is computed in place, the peers' pieces arrive by point-to-point receives;
            distances as float32 into ``wire_d`` (float32[n], root only), widened into
            ``result_d`` piece by piece.  A gather to one root uses every peer's own xGMI link
            to the root at once; a ring all-gather would push (G-1)/G of all bytes through
            every single link.
    Peers:  compute piece c into ``wire_d`` / ``wire_m`` (slice-sized) and send it while piece
            c+1 is being computed.
    Returns the list of pending communication handles, already waited on (stream-ordered
    for RCCL: the caller's current stream is made to wait, the host is not blocked).
    """
    import torch
    import torch.distributed as dist

    world, rank, root = plan.world, plan.rank, plan.root
    lo, hi = plan.bounds(rank)
    if world == 1:
        if hi > lo:
            compute(lo, hi, result_d[lo:hi], result_m[lo:hi])
        return []
    packed = wire_m is not None and wire_m.dtype == torch.uint8
    if packed and plan.align % 4 != 0:
        raise ValueError("the packed wire format needs a plan with align = 4")
    if unpack is None:
        unpack = unpack_mrca24
    pending = []
    if rank == root:
        # receives first: they only depend on the peers, so they run under the root's own kernels
        for c in range(plan.chunks):
            ops = []
            for g in range(world):
                if g == root:
                    continue
                plo, phi = plan.piece(g, c)
                if phi > plo:
                    ops.append(dist.P2POp(dist.irecv, wire_d[plo:phi], g, group))
                    ops.append(dist.P2POp(dist.irecv, wire_m[3 * plo:3 * phi] if packed else result_m[plo:phi], g, group))
            pending.append(dist.batch_isend_irecv(ops) if ops else [])
        for plo, phi in plan.pieces(rank):
            if phi > plo:
                compute(plo, phi, result_d[plo:phi], result_m[plo:phi])
        for c in range(plan.chunks):
            for w in pending[c]:
                w.wait()
            for g in range(world):
                if g == root:
                    continue
                plo, phi = plan.piece(g, c)
                if phi > plo:
                    result_d[plo:phi].copy_(wire_d[plo:phi])     # float32 -> float64, exact
                    if packed:
                        unpack(wire_m[3 * plo:3 * phi], result_m[plo:phi])
        return pending
    for c in range(plan.chunks):
        plo, phi = plan.piece(rank, c)
        if phi <= plo:
            continue
        dst_d = wire_d[plo - lo:phi - lo]
        if packed:
            at = 3 * (plo - lo)      # (a multiple of 4: plan.align)
            dst_m = wire_m[at:at + packed_bytes(phi - plo)]
            send_m = wire_m[at:at + 3 * (phi - plo)]
        else:
            dst_m = send_m = wire_m[plo - lo:phi - lo]
        compute(plo, phi, dst_d, dst_m)
        pending.append(dist.batch_isend_irecv([dist.P2POp(dist.isend, dst_d, root, group),
                                               dist.P2POp(dist.isend, send_m, root, group)]))
    for works in pending:
        for w in works:
            w.wait()     # the wire buffers may be overwritten by the next pass after this
    return pending


def run_allgather(plan: ShardPlan, compute: Callable, result_d, result_m, wire_d, wire_m, group=None,
                  unpack: Optional[Callable] = None):
    """The same pass with the result assembled on EVERY rank (``result_d`` float64[n], ``result_m`` int32[n],
    ``wire_d`` float32[n] and ``wire_m`` -- uint8 of ``packed_bytes(n)`` + slack for the packed format, or None --
    on every rank: ``sharded_buffers(plan, all_ranks=True)``).  Every rank computes piece c of its own slice in the
    wire format at the piece's global offset, sends it to every other rank and receives theirs (grouped
    point-to-point: every pair of GPUs its own xGMI link, both directions), while piece c + 1 is computed; then
    widens what it holds.  Each link carries 1/world of the result in each direction -- the per-link load of the
    gather to one root, on all links at once.  The plan must be an even split (``root_share`` None)."""
    import torch
    import torch.distributed as dist

    world, rank = plan.world, plan.rank
    if plan.root_pairs is not None:
        raise ValueError("run_allgather needs an even plan (root_share None)")
    lo, hi = plan.bounds(rank)
    if world == 1:
        if hi > lo:
            compute(lo, hi, result_d[lo:hi], result_m[lo:hi])
        return []
    packed = wire_m is not None and wire_m.dtype == torch.uint8
    if packed and plan.align % 4 != 0:
        raise ValueError("the packed wire format needs a plan with align = 4")
    if unpack is None:
        unpack = unpack_mrca24

    def ids_view(g, plo, phi, whole_dwords=False):
        """Where the ids of [plo, phi) of rank g's slice live: int32 straight in result_m, or -- packed -- 3 bytes per pair
        in g's region of wire_m, which starts on a dword of its own (the kernels store whole dwords; pieces start at
        multiples of four pairs from their slice's start: plan.align)."""
        if not packed:
            return result_m[plo:phi]
        g_lo = plan.bounds(g)[0]
        at = packed_bytes(g_lo) + 8 * g + 3 * (plo - g_lo)
        return wire_m[at:at + (packed_bytes(phi - plo) if whole_dwords else 3 * (phi - plo))]

    pending = []
    for c in range(plan.chunks):
        plo, phi = plan.piece(rank, c)
        ops = []
        if phi > plo:
            compute(plo, phi, wire_d[plo:phi], ids_view(rank, plo, phi, True))
        for g in range(world):
            if g == rank:
                continue
            glo, ghi = plan.piece(g, c)
            if ghi > glo:
                ops.append(dist.P2POp(dist.irecv, wire_d[glo:ghi], g, group))
                ops.append(dist.P2POp(dist.irecv, ids_view(g, glo, ghi), g, group))
            if phi > plo:
                ops.append(dist.P2POp(dist.isend, wire_d[plo:phi], g, group))
                ops.append(dist.P2POp(dist.isend, ids_view(rank, plo, phi), g, group))
        pending.append(dist.batch_isend_irecv(ops) if ops else [])
    for c in range(plan.chunks):
        for w in pending[c]:
            w.wait()
        for g in range(world):
            glo, ghi = plan.piece(g, c)
            if ghi > glo:
                result_d[glo:ghi].copy_(wire_d[glo:ghi])     # float32 -> float64, exact
                if packed:
                    unpack(ids_view(g, glo, ghi), result_m[glo:ghi])
    return pending


def run_local(plan: ShardPlan, compute: Callable, own_d, own_m):
    """No gather: every rank computes its slice into buffers of its own (``own_d`` float64, ``own_m`` int32, both of
    the slice's length) in ``plan.chunks`` launches; the result stays sharded over the GPUs that computed it."""
    lo, _ = plan.bounds(plan.rank)
    for plo, phi in plan.pieces(plan.rank):
        if phi > plo:
            compute(plo, phi, own_d[plo - lo:phi - lo], own_m[plo - lo:phi - lo])
    return []


def measure_root_share(world: int, rank: int, kernel_pairs_per_s: float, device=None, group=None,
                       nbytes: int = 64 << 20, root: int = 0, wire_bytes_per_pair: float = 8.0):
    """The rate at which ``root`` receives from ALL peers at once (the gather's pattern: every
    peer ships ``nbytes`` over its own link), measured on the root, then the balanced root share
    for ``kernel_pairs_per_s`` (the root's own value is used).  Collective: every rank of the
    group calls it.  Returns ``(root_share, link_bytes_per_s, kernel_pairs_per_s_of_root)``,
    identical on all ranks (broadcast by the root).  Works on RCCL (device tensors; the caller's
    current device) and gloo (CPU tensors)."""
    import time

    import torch
    import torch.distributed as dist

    def sync():
        if device is not None and torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)

    words = max(1, nbytes // 4)
    buf = torch.zeros(words * (world if rank == root else 1), dtype=torch.float32, device=device)
    link = 0.0
    for _ in range(2):                              # the first round also builds the point-to-point channels
        sync()
        dist.barrier(group)
        t0 = time.perf_counter()
        if rank == root:
            ops = [dist.P2POp(dist.irecv, buf[g * words:(g + 1) * words], g, group) for g in range(world) if g != root]
        else:
            ops = [dist.P2POp(dist.isend, buf, root, group)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        sync()
        link = 4.0 * words / max(time.perf_counter() - t0, 1e-9)      # per link, all links busy at once
    share = balanced_root_share(world, kernel_pairs_per_s, link, wire_bytes_per_pair) if rank == root else 0.0
    share = min(0.95, max(1.0 / world, share))
    t = torch.tensor([share, link, kernel_pairs_per_s], dtype=torch.float64, device=device)
    dist.broadcast(t, src=root, group=group)
    return float(t[0].item()), float(t[1].item()), float(t[2].item())


def sharded_buffers(plan: ShardPlan, device=None, packed_ids: bool = False, all_ranks: bool = False):
    """(result_d, result_m, wire_d, wire_m) torch tensors of the sizes ``run_sharded`` needs on
    this rank: results only on the root, wire buffers only where something travels.
    ``packed_ids``: MRCA ids travel as 24 bits each (``wire_m`` is uint8: on the peers 3 bytes per
    pair of the slice, on the root 3 bytes per pair of the batch, each with a few bytes of slack).
    ``all_ranks``: every rank gets the root's buffers (``run_allgather``)."""
    import torch

    lo, hi = plan.bounds(plan.rank)
    root = all_ranks or plan.rank == plan.root
    result_d = torch.empty(plan.n if root else 0, dtype=torch.float64, device=device)
    result_m = torch.empty(plan.n if root else 0, dtype=torch.int32, device=device)
    if plan.world == 1:
        return result_d, result_m, None, None
    wire_d = torch.empty(plan.n if root else hi - lo, dtype=torch.float32, device=device)
    if packed_ids:
        wire_m = torch.empty(packed_bytes(plan.n if root else hi - lo) + 16 + 8 * plan.world, dtype=torch.uint8, device=device)
    else:
        wire_m = None if root else torch.empty(hi - lo, dtype=torch.int32, device=device)
    return result_d, result_m, wire_d, wire_m


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
