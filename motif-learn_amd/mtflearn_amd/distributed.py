"""Multi-GPU sharding of the Zernike hot path: one process per GPU, one all-gather.

Every output vector depends on one K x K window only, so the path shards with no halo exchange
(SURVEY 8e): a batch of patches splits into contiguous equal blocks, a frame into contiguous
row bands (the frame itself is small and replicated).  The only collective is the all-gather
that reassembles the moment matrix on every rank -- ``torch.distributed`` with the ``nccl``
backend, which on ROCm is RCCL over xGMI; the same code runs on ``gloo`` for the CPU tests.

torch is used for device memory, streams and the collective only; the arithmetic is
``libzernike_hip.so`` called on raw device pointers.
"""
from __future__ import annotations

import numpy as np

from . import _native

__all__ = ["shard_bounds", "allgather_patch_moments", "allgather_frame_moments",
           "patch_moments_device", "frame_moments_device", "frame_maps_device",
           "sharded_patch_moments", "sharded_frame_moments"]


def shard_bounds(n_units: int, rank: int, world: int):
    """(start, count, padded): rank's contiguous block of ``n_units`` split into ``world`` blocks of
    ``padded = ceil(n_units / world)`` units; trailing ranks may own fewer (or zero) live units."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    padded = -(-n_units // world) if n_units > 0 else 0
    start = min(rank * padded, n_units)
    count = min(padded, n_units - start)
    return start, count, padded


def _current_stream_ptr(tensor):
    import torch
    return torch.cuda.current_stream(tensor.device).cuda_stream if tensor.is_cuda else 0


def patch_moments_device(plan: "_native.Plan", patches, out=None):
    """Run the batch kernel on a CUDA/HIP torch tensor ``(N, K, K)`` (float32/float64) on torch's
    current stream; returns the ``(N, n_poly)`` float64 tensor (no host copies)."""
    import torch
    assert patches.is_cuda and patches.is_contiguous()
    code = _native.ZK_F32 if patches.dtype == torch.float32 else _native.ZK_F64
    n = patches.shape[0]
    if out is None:
        out = torch.empty((n, plan.n_poly), dtype=torch.float64, device=patches.device)
    plan.transform_patches_dev(patches.data_ptr(), code, n, out.data_ptr(), _current_stream_ptr(patches))
    return out


def frame_moments_device(plan: "_native.Plan", image, row0=0, n_rows=None, out=None):
    """Run the dense kernel for output rows ``[row0, row0+n_rows)`` of a CUDA/HIP torch frame
    ``(H, W)``; returns ``(n_poly, n_rows, W)`` float64."""
    import torch
    assert image.is_cuda and image.is_contiguous()
    code = _native.ZK_F32 if image.dtype == torch.float32 else _native.ZK_F64
    h, w = image.shape
    n_rows = h - row0 if n_rows is None else n_rows
    if out is None:
        out = torch.empty((plan.n_poly, n_rows, w), dtype=torch.float64, device=image.device)
    plan.transform_frame_dev(image.data_ptr(), code, h, w, row0, n_rows, out.data_ptr(),
                             _current_stream_ptr(image))
    return out


def frame_maps_device(plan: "_native.Plan", image, n_complex, folds=(2, 3, 4, 6), m_unselect=(0, 1), p=2,
                      theta=None, want_abs=True, row0=0, n_rows=None):
    """Fused frame -> symmetry maps for output rows ``[row0, row0+n_rows)`` of a CUDA/HIP torch frame.
    Returns ``(rot, abs, mirror)`` float64 tensors of shapes ``(len(folds), n_rows, W)``,
    ``(n_complex, n_rows, W)``, ``(n_rows, W)`` (``None`` for outputs not requested).  Row bands of
    these maps are what the multi-GPU pipeline all-gathers (``allgather_frame_moments`` works on any
    ``(planes, rows, W)`` tensor): 41 planes instead of the 66 moment planes at n_max = 10."""
    import torch
    assert image.is_cuda and image.is_contiguous()
    code = _native.ZK_F32 if image.dtype == torch.float32 else _native.ZK_F64
    h, w = image.shape
    n_rows = h - row0 if n_rows is None else n_rows
    mk = lambda planes: torch.empty((planes, n_rows, w), dtype=torch.float64, device=image.device)
    rot = mk(len(folds)) if folds is not None and len(folds) else None
    ab = mk(n_complex) if want_abs else None
    mir = torch.empty((n_rows, w), dtype=torch.float64, device=image.device) if theta is not None else None
    ptr = lambda t: t.data_ptr() if t is not None else 0
    plan.frame_maps_dev(image.data_ptr(), code, h, w, row0, n_rows, folds, m_unselect, p, theta,
                        ptr(rot), ptr(ab), ptr(mir), _current_stream_ptr(image))
    return rot, ab, mir


def allgather_patch_moments(local, n_total=None, group=None, out=None):
    """All-gather equal-sized ``(padded, n_poly)`` blocks into ``(world*padded, n_poly)`` on every
    rank (one collective), trimmed to ``n_total`` rows when given."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out if n_total is None else out[:n_total]


def allgather_frame_moments(local, height=None, group=None):
    """All-gather row bands ``(n_poly, padded_rows, W)`` into the reference layout
    ``(n_poly, H, W)`` on every rank (one collective + a strided view; ``.contiguous()`` it if a
    packed array is required)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_poly, rows, width = local.shape
    slab = torch.empty((world * n_poly, rows, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(slab, local.contiguous(), group=group)
    full = slab.view(world, n_poly, rows, width).permute(1, 0, 2, 3).reshape(n_poly, world * rows, width)
    return full if height is None else full[:, :height]


def sharded_patch_moments(plan: "_native.Plan", patches, group=None):
    """Whole-job batch transform on an initialised process group: every rank passes the SAME
    ``(N, K, K)`` CUDA/HIP tensor (or at least its own block of it), computes the moments of its
    contiguous block ``shard_bounds(N, rank, world)`` and receives the full ``(N, n_poly)`` matrix from
    one all-gather.  Returns a float64 tensor on the rank's device."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = patches.shape[0]
    start, count, padded = shard_bounds(n, rank, world)
    local = torch.zeros((padded, plan.n_poly), dtype=torch.float64, device=patches.device)
    if count:
        patch_moments_device(plan, patches[start:start + count].contiguous(), out=local[:count])
    return allgather_patch_moments(local, n_total=n, group=group)


def sharded_frame_moments(plan: "_native.Plan", image, group=None):
    """Whole-job dense transform: the frame is replicated (it is small), every rank computes the row band
    ``shard_bounds(H, rank, world)`` and one all-gather reassembles ``(n_poly, H, W)`` on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    h, w = image.shape
    start, count, padded = shard_bounds(h, rank, world)
    local = torch.zeros((plan.n_poly, padded, w), dtype=torch.float64, device=image.device)
    if count:
        band = frame_moments_device(plan, image, row0=start, n_rows=count)
        local[:, :count] = band
    return allgather_frame_moments(local, height=h, group=group)
