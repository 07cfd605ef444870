"""N>1 path on CPU: world_size-2 gloo processes shard the units, 'compute' their block (the oracle
stands in for the GPU kernel here -- this test is about the sharding and the single all-gather,
not the arithmetic) and reassemble the full moment matrix with one collective."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from mtflearn_amd.distributed import shard_bounds


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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    for p in (ROOT, os.path.join(ROOT, "motif-learn_amd")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from mtflearn_amd.distributed import shard_bounds, allgather_patch_moments, allgather_frame_moments
    from oracle import zernike_oracle as zo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, m, basis = zo.zernike_basis(4, 8)
        rng = np.random.default_rng(7)                      # same data on every rank
        patches = rng.random((37, 8, 8))
        start, count, padded = shard_bounds(37, rank, world)
        block = np.zeros((padded, len(n)))
        block[:count] = zo.moments_patches(patches[start:start + count], basis)
        full = allgather_patch_moments(torch.from_numpy(block), n_total=37).numpy()
        np.save(os.path.join(tmpdir, f"patches_{rank}.npy"), full)

        frame = rng.random((21, 13))
        start, count, padded = shard_bounds(21, rank, world)
        band = np.zeros((len(n), padded, 13))
        band[:, :count] = zo.moments_frame_direct(frame, basis, rows=np.arange(start, start + count))
        fullf = allgather_frame_moments(torch.from_numpy(band), height=21).contiguous().numpy()
        np.save(os.path.join(tmpdir, f"frame_{rank}.npy"), fullf)
    finally:
        dist.destroy_process_group()


def test_two_rank_allgather_reassembles_moment_matrix(tmp_path):
    import torch.multiprocessing as mp
    from oracle import zernike_oracle as zo
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n, m, basis = zo.zernike_basis(4, 8)
    rng = np.random.default_rng(7)
    patches = rng.random((37, 8, 8))
    ref_p = zo.moments_patches(patches, basis)
    frame = rng.random((21, 13))
    ref_f = zo.moments_frame_direct(frame, basis)
    for rank in range(world):
        np.testing.assert_allclose(np.load(tmp_path / f"patches_{rank}.npy"), ref_p, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(np.load(tmp_path / f"frame_{rank}.npy"), ref_f, rtol=1e-13, atol=1e-15)
