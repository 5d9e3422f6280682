"""Device-resident use of the hot path from PyTorch: pairs already in HBM, results stay in HBM.

PyTorch is only the owner of the buffers and of the stream here; the computation is the
same C-ABI call as everywhere else (``st_distances_device`` / ``st_triangle_device``),
enqueued on torch's current stream, so it orders with the caller's other GPU work and can
feed an RCCL collective without touching the host.
"""
from typing import Optional, Tuple

from .suchtree import SuchTree


def _check_pairs(pairs):
    import torch
    if not isinstance(pairs, torch.Tensor) or not pairs.is_cuda:
        raise TypeError("pairs must be a CUDA/HIP torch tensor")
    if pairs.dtype != torch.int64:
        raise ValueError("pairs must be int64")
    if pairs.dim() != 2 or pairs.shape[1] != 2:
        raise ValueError("Expected (n, 2) tensor, got shape %s" % (tuple(pairs.shape),))
    if pairs.stride(0) < 0 or pairs.stride(1) < 0:
        raise ValueError("negative strides are not supported")


def _check_out(buf, n, dtypes, device_index, name):
    """A caller-supplied result tensor is written through its raw pointer: refuse anything that
    is not exactly an (n,) contiguous tensor of the right dtype on the tree's GPU."""
    import torch
    if not isinstance(buf, torch.Tensor) or not buf.is_cuda or buf.device.index != device_index:
        raise ValueError("%s must be a tensor on cuda:%d" % (name, device_index))
    if buf.dtype not in dtypes or buf.dim() != 1 or buf.shape[0] != n or not buf.is_contiguous():
        raise ValueError("%s must be a contiguous (%d,) tensor of dtype %s"
                         % (name, n, " or ".join(str(d) for d in dtypes)))


def distances_device(tree: SuchTree, pairs, want_dist: bool = True, want_mrca: bool = True,
                     out_dist=None, out_mrca=None, check: bool = True,
                     dist_dtype=None) -> Tuple[Optional[object], Optional[object]]:
    """(dist float64[n], mrca int32[n]) torch tensors on the tree's GPU for an int64 (n,2)
    tensor of node-id pairs (any strides).  Asynchronous on torch's current stream unless
    ``check`` (default) asks for the bounds report, which synchronises that stream and raises
    ``InvalidNodeError`` exactly like ``distances_bulk``.  ``dist_dtype=torch.float32`` returns the
    distances as the float32 values they are (the float64 form holds the same values widened)."""
    import torch
    _check_pairs(pairs)
    if dist_dtype is None:
        dist_dtype = out_dist.dtype if out_dist is not None else torch.float64
    if dist_dtype not in (torch.float64, torch.float32):
        raise ValueError("dist_dtype must be torch.float64 or torch.float32")
    dev = tree._device_tree()
    if pairs.device.index != dev.device:
        raise ValueError("pairs live on cuda:%d but the tree is on device %d" % (pairs.device.index, dev.device))
    n = int(pairs.shape[0])
    if want_dist and out_dist is None:
        out_dist = torch.empty(n, dtype=dist_dtype, device=pairs.device)
    elif want_dist:
        _check_out(out_dist, n, (torch.float64, torch.float32), dev.device, "out_dist")
    if want_mrca and out_mrca is None:
        out_mrca = torch.empty(n, dtype=torch.int32, device=pairs.device)
    elif want_mrca:
        _check_out(out_mrca, n, (torch.int32,), dev.device, "out_mrca")
    stream = torch.cuda.current_stream(pairs.device).cuda_stream
    if n:
        dev.distances_device(pairs.data_ptr(), n, out_dist.data_ptr() if want_dist else 0,
                             out_mrca.data_ptr() if want_mrca else 0, stream=stream,
                             stride0=pairs.stride(0), stride1=pairs.stride(1),
                             f32=want_dist and out_dist.dtype == torch.float32)
        if check:
            dev.fault_check(stream)
    return (out_dist if want_dist else None), (out_mrca if want_mrca else None)


def triangle_device(tree: SuchTree, ids, k_begin: int = 0, k_count: Optional[int] = None,
                    want_mrca: bool = False, check: bool = True):
    """All-pairs over a 1-D int64 id tensor: pair k = (ids[j], ids[i]), k = i(i-1)/2 + j."""
    import torch
    if not isinstance(ids, torch.Tensor) or not ids.is_cuda or ids.dtype != torch.int64 or ids.dim() != 1:
        raise TypeError("ids must be a 1-D int64 CUDA/HIP torch tensor")
    if ids.stride(0) < 0:
        raise ValueError("negative strides are not supported")
    dev = tree._device_tree()
    if ids.device.index != dev.device:
        raise ValueError("ids live on cuda:%d but the tree is on device %d" % (ids.device.index, dev.device))
    m = int(ids.shape[0])
    total = m * (m - 1) // 2
    if k_count is None:
        k_count = total - k_begin
    out_d = torch.empty(k_count, dtype=torch.float64, device=ids.device)
    out_m = torch.empty(k_count, dtype=torch.int32, device=ids.device) if want_mrca else None
    stream = torch.cuda.current_stream(ids.device).cuda_stream
    if k_count:
        dev.triangle_device(ids.data_ptr(), m, k_begin, k_count, out_d.data_ptr(),
                            out_m.data_ptr() if want_mrca else 0, stream=stream, id_stride=ids.stride(0))
        if check:
            dev.fault_check(stream)
    return out_d, out_m
