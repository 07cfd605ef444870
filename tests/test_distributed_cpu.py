"""N>1 path on CPU: world_size-2 (3, 4) gloo processes run the product's sharded drivers
(``mtflearn_amd.distributed.sharded_*``) with the test-aid communicator ``TorchComm`` -- which executes the product's
own all-gather schedule (``zk_allgather_rows_plan``) on gloo -- and an oracle-backed stand-in for the kernels.  What is under test is everything around the kernels: block and
chunk bounds (ragged tails, empty ranks), in-place placement of a rank's block inside the full result, the
windows every collective call moves, and that every rank ends with the complete, correct array.  The
arithmetic itself is the GPU suite's business (tests/test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from mtflearn_amd.distributed import _chunk_bounds, shard_bounds


def test_shard_bounds_cover_everything():
    for n_units in (0, 1, 7, 64, 1000, 4068289):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            pads = set()
            for rank in range(world):
                start, count, padded = shard_bounds(n_units, rank, world)
                assert start == seen and 0 <= count <= padded
                seen += count
                pads.add(padded)
            assert seen == n_units and len(pads) == 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_chunk_bounds_tile_the_block():
    for padded in (0, 1, 5, 64, 1000):
        for n_chunks in (1, 2, 3, 4, 7, 2000):
            cuts = _chunk_bounds(padded, n_chunks)
            assert all(b > a for a, b in cuts)
            assert [a for a, _ in cuts] == ([0] + [b for _, b in cuts][:-1] if cuts else [])
            assert (cuts[-1][1] if cuts else 0) == padded and len(cuts) <= max(1, n_chunks)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleCompute:
    """Stand-in for ``DeviceCompute``: same four methods, arithmetic by the CPU oracle on CPU tensors.
    Result arrays start as NaN so that any element no rank or collective wrote shows up."""

    def __init__(self, n_max, size):
        from oracle import zernike_oracle as zo
        self.zo = zo
        self.n, self.m, self.basis = zo.zernike_basis(n_max, size)
        self.n_poly = len(self.n)

    def empty(self, shape, like):
        import torch
        return torch.full(tuple(shape), float("nan"), dtype=torch.float64)

    def stream(self, tensor):
        return 0

    def patches(self, patches, out_rows):
        import torch
        out_rows.copy_(torch.from_numpy(self.zo.moments_patches(patches.numpy(), self.basis)))

    def frame_band(self, image, row0, n_rows, full):
        import torch
        band = self.zo.moments_frame_direct(image.numpy(), self.basis, rows=np.arange(row0, row0 + n_rows))
        full[:, row0:row0 + n_rows] = torch.from_numpy(band)

    def frame(self, image, out):
        import torch
        out.copy_(torch.from_numpy(self.zo.moments_frame_direct(image.numpy(), self.basis)))

    def maps_band(self, image, row0, n_rows, full, n_complex, folds, m_unselect, p, theta):
        import torch
        zo = self.zo
        band = zo.moments_frame_direct(image.numpy(), self.basis, rows=np.arange(row0, row0 + n_rows))
        rot, ab, mir = full
        if rot is not None:
            rot[:, row0:row0 + n_rows] = torch.from_numpy(zo.rot_maps(band, self.n, self.m, list(folds), p=p, m_unselect=m_unselect))
        if ab is not None:
            ab[:, row0:row0 + n_rows] = torch.from_numpy(np.abs(zo.to_complex(band, self.n, self.m)[0]))
        if mir is not None:
            mir[row0:row0 + n_rows] = torch.from_numpy(zo.mirror_map(band, self.n, self.m, theta=theta, p=p, m_unselect=m_unselect))


def _inputs():
    rng = np.random.default_rng(7)                      # same data on every rank
    return rng.random((37, 8, 8)), rng.random((21, 13)), rng.random((5, 12, 11))


THETA = np.linspace(0, 2 * np.pi, 24, endpoint=False)


def _worker(rank, world, port, tmpdir, algo):
    for p in (ROOT, os.path.join(ROOT, "motif-learn_amd")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from mtflearn_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = D.TorchComm(algo=algo)
        oc = _OracleCompute(4, 8)
        patches, frame, frames = _inputs()
        n_c = sum(n // 2 + 1 for n in range(5))
        for n_chunks in (1, 3):
            start, count, _ = D.shard_bounds(37, rank, world)
            full = D.sharded_patch_moments(oc, comm, torch.from_numpy(patches[start:start + count]), 37, n_chunks=n_chunks)
            np.save(os.path.join(tmpdir, f"patches_{n_chunks}_{rank}.npy"), full.numpy())
            fullf = D.sharded_frame_moments(oc, comm, torch.from_numpy(frame), n_chunks=n_chunks)
            np.save(os.path.join(tmpdir, f"frame_{n_chunks}_{rank}.npy"), fullf.numpy())
            rot, ab, mir = D.sharded_frame_maps(oc, comm, torch.from_numpy(frame), n_c, folds=(2, 3), theta=THETA,
                                                n_chunks=n_chunks)
            np.savez(os.path.join(tmpdir, f"maps_{n_chunks}_{rank}.npz"), rot=rot.numpy(), ab=ab.numpy(), mir=mir.numpy())
        start, count, _ = D.shard_bounds(5, rank, world)
        fullb = D.sharded_frames_moments(oc, comm, torch.from_numpy(frames[start:start + count]), 5)
        np.save(os.path.join(tmpdir, f"frames_{rank}.npy"), fullb.numpy())
        assert comm.calls > 0                                # the planner's entries were what moved the data
        # the timing reduction of bench.py
        assert comm.max_over_ranks(float(rank)) == float(world - 1)
        comm.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,algo", [(2, "auto"), (3, "auto"), (2, "bcast"), (3, "p2p"), (4, "allgather")])
def test_sharded_drivers_reassemble_everything(tmp_path, world, algo):
    """The drivers at world 2 - 4 on gloo.  TorchComm executes the list of zk_allgather_rows_plan -- the schedule the
    RCCL executor runs -- so the windows, the pairing and the order of the shipped collective are what is tested."""
    import torch.multiprocessing as mp
    from oracle import zernike_oracle as zo
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), algo), nprocs=world, join=True)
    n, m, basis = zo.zernike_basis(4, 8)
    patches, frame, frames = _inputs()
    ref_p = zo.moments_patches(patches, basis)
    ref_f = zo.moments_frame_direct(frame, basis)
    ref_rot = zo.rot_maps(ref_f, n, m, [2, 3], p=2, m_unselect=(0, 1))
    ref_abs = np.abs(zo.to_complex(ref_f, n, m)[0])
    ref_mir = zo.mirror_map(ref_f, n, m, theta=THETA, p=2, m_unselect=(0, 1))
    ref_b = np.stack([zo.moments_frame_direct(f, basis) for f in frames])
    tol = dict(rtol=1e-12, atol=1e-14)
    for rank in range(world):
        for n_chunks in (1, 3):
            np.testing.assert_allclose(np.load(tmp_path / f"patches_{n_chunks}_{rank}.npy"), ref_p, **tol)
            np.testing.assert_allclose(np.load(tmp_path / f"frame_{n_chunks}_{rank}.npy"), ref_f, **tol)
            with np.load(tmp_path / f"maps_{n_chunks}_{rank}.npz") as f:
                np.testing.assert_allclose(f["rot"], ref_rot, **tol)
                np.testing.assert_allclose(f["ab"], ref_abs, **tol)
                np.testing.assert_allclose(f["mir"], ref_mir, **tol)
        np.testing.assert_allclose(np.load(tmp_path / f"frames_{rank}.npy"), ref_b, **tol)


def test_device_helpers_reject_foreign_dtypes_and_devices():
    """a float16 / integer tensor must not be read as float64, a CPU tensor must not reach a kernel."""
    import torch
    from mtflearn_amd import distributed as D

    class _Plan:
        device, n_poly = 0, 3

    with pytest.raises(ValueError, match="must live on the GPU"):
        D.patch_moments_device(_Plan(), torch.zeros((2, 8, 8)))
    for dt in (torch.float16, torch.bfloat16, torch.int16, torch.uint8):
        with pytest.raises(TypeError, match="float32 or float64"):
            D._dtype_code(torch.zeros(4, dtype=dt))
    assert D._dtype_code(torch.zeros(1, dtype=torch.float32)) == 0 and D._dtype_code(torch.zeros(1, dtype=torch.float64)) == 1
