"""Multi-GPU sharding of the Zernike hot path: one process per GPU, one exchange step.

Every output vector depends on one K x K window only, so the path shards with no halo exchange and no
data-path collective: a batch of patches splits into contiguous equal blocks
(:func:`shard_bounds`), a frame into contiguous row bands (the frame itself is small and replicated), a batch
of frames into whole frames.  The only exchange is the all-gather that reassembles the result on every rank.

Layout is chosen so that the gather needs no copy on either side: every rank holds the FULL result array
(``(N, n_poly)``, ``(n_poly, H, W)``, ``(planes, H, W)`` maps or ``(F, n_poly, H, W)``), its kernels write
its own block straight into place (dense kernels through their plane stride), and the collective fills in the
other blocks in place (``zk_allgather_rows``).  The drivers below cut a rank's block into chunks and issue
kernel(c+1) while the transfer of chunk c runs on the communicator's own stream.

Two communicators share one interface:

* :class:`RcclComm` -- the product: ``zk_comm_*`` / ``zk_allgather_rows`` of ``libzernike_hip.so`` on RCCL
  (no ``torch.distributed`` involved; rendezvous through a file, a TCP port, or an id the caller moved).
* :class:`TorchComm` -- a test aid on ``torch.distributed`` (``gloo`` on CPU for the world-size-2 tests and
  for rehearsing the control flow with two ranks on one GPU).

torch is used for device memory and streams only; the arithmetic is ``libzernike_hip.so`` called on raw
device pointers.
"""
from __future__ import annotations

import struct

import numpy as np

from . import _native

__all__ = ["shard_bounds", "one_gpu_rank_env", "RcclComm", "TorchComm", "DeviceCompute", "patch_moments_device",
           "frame_moments_device", "frame_maps_device", "sharded_patch_moments", "sharded_frame_moments",
           "sharded_frame_maps", "sharded_frames_moments"]


def shard_bounds(n_units: int, rank: int, world: int):
    """(start, count, padded): rank's contiguous block of ``n_units`` split into ``world`` blocks of
    ``padded = ceil(n_units / world)`` units; trailing ranks may own fewer (or zero) live units."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    padded = -(-n_units // world) if n_units > 0 else 0
    start = min(rank * padded, n_units)
    count = min(padded, n_units - start)
    return start, count, padded


def one_gpu_rank_env(rank, base=None):
    """Environment for rank ``rank`` of a REHEARSAL of several RCCL ranks on ONE GPU (a test aid: tests/
    test_gpu_multirank.py, ``bench.py --gpus N`` with ``ZK_BENCH_ONE_DEVICE=1``).  RCCL refuses two ranks of one
    communicator on the same device of the same host ("Duplicate GPU detected"); with a different ``NCCL_HOSTID`` per
    process it takes the ranks for different hosts and connects them through its socket transport on the loopback
    interface.  The data then moves over TCP instead of xGMI -- timings mean nothing -- but every call of the shipped
    collective (``ncclSend`` / ``ncclRecv`` groups, ``ncclAllGather``, ``ncclBroadcast`` as ``zk_allgather_rows``
    issues them) executes in librccl with more than one rank."""
    import os
    env = dict(os.environ if base is None else base)
    env.update(NCCL_HOSTID=f"zk-one-gpu-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
               NCCL_SHM_DISABLE="1", NCCL_P2P_DISABLE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def rccl_debug_env(rank, base=None, directory="/tmp"):
    """Environment additions that make RCCL write, per rank, which transport it connected every channel over
    (``NCCL_DEBUG=INFO`` restricted to the connection subsystems, into a file of this rank's own): what
    :func:`rccl_transport_summary` reads afterwards.  A level below INFO (unset, ``VERSION``, ``WARN`` -- the image exports
    ``VERSION``) is raised to INFO; a caller's ``INFO`` / ``TRACE`` run, its subsystem list and its log file are left alone."""
    import os
    env = {}
    have = os.environ if base is None else base
    if have.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        env["NCCL_DEBUG"] = "INFO"
        if "NCCL_DEBUG_SUBSYS" not in have:
            env["NCCL_DEBUG_SUBSYS"] = "INIT,P2P,NET,SHM"
    if "NCCL_DEBUG_FILE" not in have:
        env["NCCL_DEBUG_FILE"] = os.path.join(directory, f"zk_rccl_{os.getpid()}_rank{rank}.log")
    return env


def rccl_transport_summary(log_text):
    """Channels per transport from an RCCL INFO log: the ``Channel NN : a[dev] -> b[dev] via P2P/IPC`` /
    ``... [send] via NET/Socket/0`` lines librccl prints when it connects a channel.  Returns ``{transport: count}``, e.g.
    ``{"P2P/IPC": 112}`` on an xGMI node or ``{"NET/Socket": 24}`` for ranks that met over sockets."""
    import re
    counts = {}
    for m in re.finditer(r"Channel \d+(?:/\d+)? *: *\d+\[[^\]]*\] *-> *\d+\[[^\]]*\](?: \[(?:send|receive)\])? via ([A-Za-z0-9]+(?:/[A-Za-z]+)?)",
                         log_text):
        counts[m.group(1)] = counts.get(m.group(1), 0) + 1
    return counts


def describe_transport(per_rank_counts, rehearsal=False):
    """One string for the bench line from every rank's ``rccl_transport_summary``: the transports seen, most used first --
    ``"P2P/IPC"`` when every channel of every rank is peer-to-peer (xGMI inside a node), ``"NET/Socket (rehearsal)"``
    for the one-GPU rehearsal, ``"unknown (no channel lines in the RCCL log)"`` when nothing could be parsed."""
    total = {}
    for c in per_rank_counts:
        for k, v in c.items():
            total[k] = total.get(k, 0) + v
    if not total:
        return "unknown (no channel lines in the RCCL log)"
    names = " + ".join(k for k, _ in sorted(total.items(), key=lambda kv: -kv[1]))
    return names + (" (rehearsal)" if rehearsal else "")


def _chunk_bounds(padded: int, n_chunks: int):
    """Cut ``[0, padded)`` into at most ``n_chunks`` consecutive non-empty windows (same on every rank)."""
    n_chunks = max(1, min(int(n_chunks), padded)) if padded > 0 else 1
    edges = [padded * c // n_chunks for c in range(n_chunks + 1)]
    return [(edges[c], edges[c + 1]) for c in range(n_chunks) if edges[c + 1] > edges[c]]


def _is_native(array):
    """A :class:`mtflearn_amd._native.DeviceArray` (device memory of the library's own, no torch involved)?"""
    return isinstance(array, _native.DeviceArray)


def _empty_like(shape, like):
    """Uninitialised float64 device array of ``shape`` on ``like``'s device, of ``like``'s kind."""
    if _is_native(like):
        return _native.DeviceArray(shape, np.float64, like.device.index)
    import torch
    return torch.empty(shape, dtype=torch.float64, device=like.device)


def _current_stream_ptr(tensor):
    if _is_native(tensor):
        return 0                                  # the device's default stream
    import torch
    return torch.cuda.current_stream(tensor.device).cuda_stream if tensor.is_cuda else 0


def _is_f64(array):
    if _is_native(array):
        return array.dtype == np.float64
    import torch
    return array.dtype == torch.float64


def _dtype_code(tensor):
    """ZK_F32 / ZK_F64 of a torch tensor or DeviceArray; anything else is an error (the kernels would read it as float64)."""
    if _is_native(tensor):
        if tensor.dtype == np.float32:
            return _native.ZK_F32
        if tensor.dtype == np.float64:
            return _native.ZK_F64
        raise TypeError(f"the device entry points take float32 or float64 arrays, not {tensor.dtype}; convert first "
                        "(ZPs.transform does that for NumPy input)")
    import torch
    if tensor.dtype == torch.float32:
        return _native.ZK_F32
    if tensor.dtype == torch.float64:
        return _native.ZK_F64
    raise TypeError(f"the device entry points take float32 or float64 tensors, not {tensor.dtype}; convert first "
                    "(ZPs.transform does that for NumPy input)")


def _check_operand(plan, tensor, what):
    if not tensor.is_cuda:
        raise ValueError(f"{what} must live on the GPU")
    if not tensor.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    if tensor.device.index != plan.device:
        raise ValueError(f"{what} is on cuda:{tensor.device.index} but the plan was created on device {plan.device}: "
                         "create the plan on the tensor's device (MTFLEARN_AMD_DEVICE / ZPs.to_device)")


# ---------------------------------------------------------------------------------------------------------
# single-GPU device entry points on torch tensors
# ---------------------------------------------------------------------------------------------------------
def patch_moments_device(plan: "_native.Plan", patches, out=None):
    """Run the batch kernel on a CUDA/HIP torch tensor ``(N, K, K)`` (float32/float64) on torch's
    current stream; returns the ``(N, n_poly)`` float64 tensor (no host copies)."""
    _check_operand(plan, patches, "patches")
    code = _dtype_code(patches)
    n = patches.shape[0]
    if out is None:
        out = _empty_like((n, plan.n_poly), patches)
    else:
        _check_operand(plan, out, "out")
    plan.transform_patches_dev(patches.data_ptr(), code, n, out.data_ptr(), _current_stream_ptr(patches))
    return out


def frame_moments_device(plan: "_native.Plan", image, row0=0, n_rows=None, out=None, full=None):
    """Run the dense kernel for output rows ``[row0, row0+n_rows)`` of a CUDA/HIP torch frame ``(H, W)``.
    Returns ``(n_poly, n_rows, W)`` float64 -- or, with ``full`` (an ``(n_poly, H, W)`` tensor), writes the
    band in place into it and returns ``full``."""
    _check_operand(plan, image, "image")
    code = _dtype_code(image)
    h, w = image.shape
    n_rows = h - row0 if n_rows is None else n_rows
    stream = _current_stream_ptr(image)
    if full is not None:
        _check_operand(plan, full, "full")
        assert tuple(full.shape) == (plan.n_poly, h, w) and _is_f64(full)
        plan.transform_frame_dev(image.data_ptr(), code, h, w, row0, n_rows, full.data_ptr() + row0 * w * 8, stream,
                                 plane_stride=h * w)
        return full
    if out is None:
        out = _empty_like((plan.n_poly, n_rows, w), image)
    else:
        _check_operand(plan, out, "out")
    plan.transform_frame_dev(image.data_ptr(), code, h, w, row0, n_rows, out.data_ptr(), stream)
    return out


def frame_maps_device(plan: "_native.Plan", image, n_complex, folds=(2, 3, 4, 6), m_unselect=(0, 1), p=2,
                      theta=None, want_abs=True, row0=0, n_rows=None, full=None):
    """Fused frame -> symmetry maps for output rows ``[row0, row0+n_rows)`` of a CUDA/HIP torch frame.
    Returns ``(rot, abs, mirror)`` float64 tensors of shapes ``(len(folds), n_rows, W)``,
    ``(n_complex, n_rows, W)``, ``(n_rows, W)`` (``None`` for outputs not requested).  With
    ``full = (rot_full, abs_full, mirror_full)`` (whole-frame tensors, entries ``None`` where not wanted) the
    band is written in place into them and ``full`` is returned."""
    _check_operand(plan, image, "image")
    code = _dtype_code(image)
    h, w = image.shape
    n_rows = h - row0 if n_rows is None else n_rows
    stream = _current_stream_ptr(image)
    if full is not None:
        ptr = lambda t: t.data_ptr() + row0 * w * 8 if t is not None else 0
        plan.frame_maps_dev(image.data_ptr(), code, h, w, row0, n_rows, folds if full[0] is not None else None,
                            m_unselect, p, theta if full[2] is not None else None, ptr(full[0]), ptr(full[1]),
                            ptr(full[2]), stream, plane_stride=h * w)
        return full
    mk = lambda planes: _empty_like((planes, n_rows, w), image)
    rot = mk(len(folds)) if folds is not None and len(folds) else None
    ab = mk(n_complex) if want_abs else None
    mir = _empty_like((n_rows, w), image) if theta is not None else None
    ptr = lambda t: t.data_ptr() if t is not None else 0
    plan.frame_maps_dev(image.data_ptr(), code, h, w, row0, n_rows, folds, m_unselect, p, theta,
                        ptr(rot), ptr(ab), ptr(mir), stream)
    return rot, ab, mir


# ---------------------------------------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------------------------------------
class RcclComm:
    """This process's endpoint of the RCCL communicator inside ``libzernike_hip.so`` (``zk_comm_*``).

    ``RcclComm(device, rank, world, path=...)`` (ranks of one node meet through a file), ``port=`` (TCP) or
    ``unique_id=``; see :class:`mtflearn_amd._native.Comm`."""

    def __init__(self, device, rank, world, **rendezvous):
        self._c = _native.Comm(device, rank, world, **rendezvous)
        self.rank, self.world, self.device = self._c.rank, self._c.world, self._c.device

    def allgather_rows(self, full, n_planes, height, width, rows_per_rank, row_off, n_rows, stream=0):
        if not full.is_cuda or full.device.index != self.device or not full.is_contiguous():
            raise ValueError("the gathered array must be a contiguous tensor on the communicator's device")
        if full.element_size() != 8 or full.numel() != n_planes * height * width:
            raise ValueError("the gathered array must hold n_planes * height * width float64 values")
        self._c.allgather_rows(full.data_ptr(), n_planes, height, width, rows_per_rank, row_off, n_rows, stream)

    def join(self, stream=0):
        self._c.join(stream)

    def ranks_seen(self):
        return self._c.ranks_seen()

    def allgather_host(self, payload: bytes):
        return self._c.allgather_host(payload)

    def max_over_ranks(self, value: float) -> float:
        return max(struct.unpack("d", b)[0] for b in self.allgather_host(struct.pack("d", float(value))))

    def barrier(self):
        self.allgather_host(b"\0")

    def close(self):
        self._c.close()


class TorchComm:
    """Same interface on ``torch.distributed`` -- a TEST AID (``gloo`` on CPU tensors for the world-size-2 / 3
    tests, ``gloo`` with GPU tensors staged through the host to rehearse several ranks on one GPU).  Blocking;
    ``stream`` is ignored.  ``allgather_rows`` executes the SAME schedule as the product: the list
    ``zk_allgather_rows_plan`` returns for this rank (the planner half of ``zk_allgather_rows``), entry by entry,
    with ``isend`` / ``irecv`` / ``broadcast`` / ``all_gather_into_tensor`` standing in for the RCCL calls and a
    wait at every group boundary -- so an error in the window arithmetic, in the pairing of sends and receives or
    in their order (gloo, like RCCL, matches point-to-point messages between two ranks in issue order) fails the
    gloo tests.  ``algo``: ``"p2p" | "allgather" | "bcast"`` as ``ZK_COMM_ALGO`` (default: the environment's)."""

    def __init__(self, group=None, algo=None):
        import os
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.algo = _native.COMM_ALGOS.get(os.environ.get("ZK_COMM_ALGO", "") if algo is None else algo, _native.COMM_AUTO)
        self.calls = 0          # RCCL-call stand-ins executed so far (tests read it)

    def _global(self, r):
        return r if self._group is None else self._dist.get_global_rank(self._group, r)

    def allgather_rows(self, full, n_planes, height, width, rows_per_rank, row_off, n_rows, stream=0):
        import torch
        dist = self._dist
        plan = _native.allgather_rows_plan(self.rank, self.world, n_planes, height, width, rows_per_rank, row_off,
                                           n_rows, self.algo)
        if not plan:
            return
        if full.numel() != n_planes * height * width or not full.is_contiguous():
            raise ValueError("the gathered array must be contiguous and hold n_planes * height * width values")
        on_gpu = full.is_cuda
        if on_gpu:
            torch.cuda.current_stream(full.device).synchronize()
        flat = full.view(-1)
        pending, landed = [], []       # requests of the open group; (host buffer, offset) pairs to copy back to the GPU

        def run(seg):                  # the tensor gloo works on: the run itself (CPU) or a host copy of it (GPU)
            return seg.cpu() if on_gpu else seg

        def flush():
            for req in pending:
                req.wait()
            pending.clear()
            for buf, off in landed:
                flat[off:off + buf.numel()].copy_(buf)
            landed.clear()

        group_now = plan[0][2]
        for op, peer, group, _plane, off, cnt in plan:
            if group != group_now:
                flush()
                group_now = group
            seg = flat[off:off + cnt]
            self.calls += 1
            if op == _native.XFER_SEND:
                pending.append(dist.isend(run(seg), self._global(peer), group=self._group))
            elif op == _native.XFER_RECV:
                buf = torch.empty(cnt, dtype=full.dtype) if on_gpu else seg
                pending.append(dist.irecv(buf, self._global(peer), group=self._group))
                if on_gpu:
                    landed.append((buf, off))
            elif op == _native.XFER_BCAST:
                flush()                # a collective does not overtake the point-to-point calls issued before it
                buf = run(seg)
                dist.broadcast(buf, src=self._global(peer), group=self._group)
                if on_gpu and peer != self.rank:
                    seg.copy_(buf)
            else:                      # XFER_ALLGATHER: send = own block, receive = the run of all blocks around it
                flush()
                base = off - self.rank * cnt
                out = torch.empty(cnt * self.world, dtype=full.dtype)
                dist.all_gather_into_tensor(out, run(seg).clone(), group=self._group)
                flat[base:base + cnt * self.world].copy_(out)
        flush()

    def join(self, stream=0):
        pass

    def allgather_host(self, payload: bytes):
        out = [None] * self.world
        self._dist.all_gather_object(out, payload, group=self._group)
        return out

    def max_over_ranks(self, value: float) -> float:
        return max(struct.unpack("d", b)[0] for b in self.allgather_host(struct.pack("d", float(value))))

    def barrier(self):
        self._dist.barrier(group=self._group)

    def close(self):
        pass


# ---------------------------------------------------------------------------------------------------------
# what a driver asks of "the kernels": the product adapter calls libzernike_hip.so; the CPU tests plug in
# an oracle-backed stand-in with the same four methods (tests/test_distributed_cpu.py)
# ---------------------------------------------------------------------------------------------------------
class DeviceCompute:
    def __init__(self, plan: "_native.Plan"):
        self.plan, self.n_poly = plan, plan.n_poly

    def empty(self, shape, like):
        return _empty_like(shape, like)

    def stream(self, tensor):
        return _current_stream_ptr(tensor)

    def patches(self, patches, out_rows):
        patch_moments_device(self.plan, patches, out=out_rows)

    def frame_band(self, image, row0, n_rows, full):
        frame_moments_device(self.plan, image, row0=row0, n_rows=n_rows, full=full)

    def frame(self, image, out):
        frame_moments_device(self.plan, image, out=out)

    def maps_band(self, image, row0, n_rows, full, n_complex, folds, m_unselect, p, theta):
        frame_maps_device(self.plan, image, n_complex, folds=folds, m_unselect=m_unselect, p=p, theta=theta,
                          row0=row0, n_rows=n_rows, full=full)


def _as_compute(plan_or_compute):
    return plan_or_compute if hasattr(plan_or_compute, "frame_band") else DeviceCompute(plan_or_compute)


# ---------------------------------------------------------------------------------------------------------
# sharded drivers: kernel(chunk c+1) overlaps the transfer of chunk c; every rank ends with the full result
# ---------------------------------------------------------------------------------------------------------
def sharded_patch_moments(plan, comm, patches_local, n_total, out=None, n_chunks=4):
    """Batch of ``n_total`` patches split into the blocks of :func:`shard_bounds`; ``patches_local`` is this
    rank's block ``(count, K, K)``.  Returns the full ``(n_total, n_poly)`` float64 matrix on every rank."""
    compute = _as_compute(plan)
    start, count, padded = shard_bounds(n_total, comm.rank, comm.world)
    if patches_local.shape[0] != count:
        raise ValueError(f"rank {comm.rank} owns {count} of {n_total} patches, got {patches_local.shape[0]}")
    full = compute.empty((n_total, compute.n_poly), patches_local) if out is None else out
    stream = compute.stream(full)
    for c0, c1 in _chunk_bounds(padded, n_chunks):
        lo, hi = min(c0, count), min(c1, count)
        if hi > lo:
            compute.patches(patches_local[lo:hi], full[start + lo:start + hi])
        comm.allgather_rows(full, 1, n_total, compute.n_poly, padded, c0, c1 - c0, stream)
    comm.join(stream)
    return full


def sharded_frame_moments(plan, comm, image, out=None, n_chunks=4):
    """Dense transform of one frame (replicated on every rank -- it is small): rank r computes the row band
    ``shard_bounds(H, r, world)`` in place into the full ``(n_poly, H, W)`` array, sub-band by sub-band, each
    gathered while the next one is computed."""
    compute = _as_compute(plan)
    h, w = image.shape
    start, count, padded = shard_bounds(h, comm.rank, comm.world)
    full = compute.empty((compute.n_poly, h, w), image) if out is None else out
    stream = compute.stream(full)
    for c0, c1 in _chunk_bounds(padded, n_chunks):
        lo, hi = min(c0, count), min(c1, count)
        if hi > lo:
            compute.frame_band(image, start + lo, hi - lo, full)
        comm.allgather_rows(full, compute.n_poly, h, w, padded, c0, c1 - c0, stream)
    comm.join(stream)
    return full


def sharded_frame_maps(plan, comm, image, n_complex, folds=(2, 3, 4, 6), m_unselect=(0, 1), p=2, theta=None,
                       want_abs=True, n_chunks=4, out=None):
    """configs[4]: frame -> fused symmetry maps, row bands sharded, the maps (not the moments) gathered:
    ``len(folds) + n_complex + 1`` planes instead of ``n_poly``.  Returns ``(rot, abs, mirror)`` whole-frame
    tensors (``None`` where not requested) on every rank; ``out = (rot, abs, mirror)`` supplies them."""
    compute = _as_compute(plan)
    h, w = image.shape
    start, count, padded = shard_bounds(h, comm.rank, comm.world)
    n_folds = len(folds) if folds is not None else 0
    if out is not None:
        rot, ab, mir = out
    else:
        rot = compute.empty((n_folds, h, w), image) if n_folds else None
        ab = compute.empty((n_complex, h, w), image) if want_abs else None
        mir = compute.empty((h, w), image) if theta is not None else None
    full = (rot, ab, mir)
    stream = compute.stream(image)
    for c0, c1 in _chunk_bounds(padded, n_chunks):
        lo, hi = min(c0, count), min(c1, count)
        if hi > lo:
            compute.maps_band(image, start + lo, hi - lo, full, n_complex, folds, m_unselect, p, theta)
        for t in full:
            if t is not None:
                comm.allgather_rows(t, t.numel() // (h * w), h, w, padded, c0, c1 - c0, stream)
    comm.join(stream)
    return full


def sharded_frames_moments(plan, comm, frames_local, n_frames, out=None):
    """configs[3]: a batch of ``n_frames`` frames sharded by whole frames (``frames_local``: this rank's
    ``(count, H, W)`` block).  Frame i of every rank is transformed, then gathered while frame i+1 is
    computed; returns the full ``(n_frames, n_poly, H, W)`` array on every rank."""
    compute = _as_compute(plan)
    start, count, padded = shard_bounds(n_frames, comm.rank, comm.world)
    if frames_local.shape[0] != count:
        raise ValueError(f"rank {comm.rank} owns {count} of {n_frames} frames, got {frames_local.shape[0]}")
    h, w = frames_local.shape[1:]
    full = compute.empty((n_frames, compute.n_poly, h, w), frames_local) if out is None else out
    stream = compute.stream(full)
    for i in range(padded):
        if i < count:
            compute.frame(frames_local[i], full[start + i])
        comm.allgather_rows(full, 1, n_frames, compute.n_poly * h * w, padded, i, 1, stream)
    comm.join(stream)
    return full
