"""Round-2 runtime around the kernels, on the GPU through the C ABI: the chunked host pipeline and the
page-locked result pool, the plane-stride (in-place band) entry points, launch splitting of very tall bands,
the symmetry-map option-table cache, and the RCCL communicator of libzernike_hip.so at world size 1 (the
1-GPU box cannot host two RCCL ranks; world 2 / 3 control flow runs on gloo in tests/test_distributed_cpu.py)."""
import os
import warnings

import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu


def _zps(n_max, size):
    from mtflearn_amd import ZPs
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ZPs(n_max, size)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


# ------------------------------------------------------------------------------------------------ host pipeline
def test_host_pipeline_many_chunks_equals_one_chunk():
    """ZPs.transform with NumPy buffers: a job cut into dozens of chunks (ring slots reused many times) gives
    bit-identical results to the same job in one chunk, for every host entry point."""
    from oracle import zernike_oracle as zo
    rng = np.random.default_rng(21)
    z = _zps(8, 32)
    plan = z._device_plan()
    patches = rng.random((5000, 32, 32), dtype=np.float32)
    frame = rng.random((300, 420), dtype=np.float32)
    pts = np.column_stack([rng.integers(0, 420, 3000), rng.integers(0, 300, 3000)])
    theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)

    def run():
        zp = plan.transform_patches(patches)
        zf = plan.transform_frame(frame)
        zk = plan.transform_points(frame, pts)
        maps = plan.frame_maps(frame, 25, folds=[2, 3, 4, 6], theta=theta)
        return zp, zf, zk, maps

    plan.set_host_chunk(1 << 30)
    one = run()
    plan.set_host_chunk(1 << 20)          # 1 MiB: ~235 patches -> rounded to 256 per chunk -> 20 chunks; 8-row bands
    many = run()
    plan.set_host_chunk(0)
    # dense results do not depend on the cut (same kernel, one lane per output pixel): bit-identical
    np.testing.assert_array_equal(one[1], many[1])
    for a, b in zip(one[3], many[3]):
        np.testing.assert_array_equal(a, b)
    # batch / key-point results may come from another kernel variant at another batch size (ZK_PATH_AUTO looks at
    # the count): equal to rounding
    for k in (0, 2):
        np.testing.assert_allclose(many[k], one[k], rtol=0, atol=1e-13 * np.abs(one[k]).max())
    rel_close(many[0], zo.moments_patches(patches, z.polynomials))
    rel_close(many[1], zo.moments_frame_direct(frame, z.polynomials))
    z.release()
    np.testing.assert_array_equal(z.transform(patches).data, one[0])        # staging is re-created on demand


def test_pinned_result_pool_recycles_blocks():
    from mtflearn_amd import _native
    pool = _native.pinned
    pool.trim()
    z = _zps(8, 32)
    frame = np.random.default_rng(3).random((256, 256), dtype=np.float32)
    a = z.transform(frame).data                                               # 45 x 256 x 256 x 8 = 23.6 MB: pooled
    assert a.flags.c_contiguous and a.dtype == np.float64 and not a.flags.owndata
    addr = a.ctypes.data
    keep = a.copy()
    del a
    assert len(pool._free) == 1
    b = z.transform(frame).data
    assert b.ctypes.data == addr and len(pool._free) == 0                     # the block came back
    np.testing.assert_array_equal(b, keep)
    view = b[3:5]
    del b
    assert len(pool._free) == 0                                               # a live view keeps the block out
    del view
    assert len(pool._free) == 1
    pool.trim()
    assert len(pool._free) == 0
    small = z.transform(np.zeros((4, 32, 32), np.float32)).data
    assert small.flags.owndata                                                # small results are plain arrays


def test_user_constructed_container_copies_like_the_reference():
    from mtflearn_amd import zmoments
    z = _zps(4, 8)
    data = np.arange(2 * 15, dtype=np.float64).reshape(2, 15)
    zm = zmoments(data, z.n, z.m)
    zm.data[0, 0] = -1
    assert data[0, 0] == 0                                                    # reference _zmoments.py:277 always copies


# ------------------------------------------------------------------------------------------------ strided bands
def test_tall_band_is_split_into_launches():
    """More rows than one grid can cover (65535 blocks of 4 rows): the launchers cut the band."""
    torch = _torch()
    from mtflearn_amd.distributed import frame_moments_device
    z = _zps(10, 12)
    plan = z._device_plan()
    H, W = 262140 + 133, 24
    img = torch.rand((H, W), device="cuda")
    full = frame_moments_device(plan, img)
    assert torch.isfinite(full).all()
    lo = frame_moments_device(plan, img, row0=0, n_rows=200000)
    hi = frame_moments_device(plan, img, row0=200000, n_rows=H - 200000)
    assert torch.equal(torch.cat([lo, hi], dim=1), full)


def test_maps_option_tables_are_cached_and_evicted():
    """More distinct (folds, m_unselect, theta) option sets than the plan caches: every call still gets ITS
    table (compared with the container's tail on the device moments), repeated sets hit the cache."""
    z = _zps(8, 16)
    frame = np.random.default_rng(8).random((40, 52))
    zm = z.transform(frame)
    sets = [((2, 3), (0, 1), 12), ((4,), (0,), 8), ((2, 6), (0, 1, 2), 16), ((3,), (0, 1), 360), ((5, 2), (0, 3), 20),
            ((2,), (0, 1), 24), ((2, 3, 4, 6), (0, 1), 360), ((7,), (0,), 4), ((2, 4), (0, 1), 40), ((6,), (0, 2), 28)]
    for rep in range(2):
        for folds, unsel, nth in sets + sets[:3]:
            theta = np.linspace(0, 2 * np.pi, nth, endpoint=False)
            out = z.symmetry_maps(frame, n_folds=folds, m_unselect=unsel, theta=theta)
            rel_close(out["rot_maps"], zm.rot_maps(list(folds), m_unselect=unsel), rtol=1e-9, atol_scale=1e-11)
            rel_close(out["mirror_map"], zm.mirror_map(theta=theta, m_unselect=unsel), rtol=1e-9, atol_scale=1e-11)


# ------------------------------------------------------------------------------------------------ communicator
@pytest.mark.parametrize("rendezvous", ["file", "tcp", "id"])
def test_rccl_communicator_world_one(tmp_path, rendezvous):
    """libzernike_hip.so's own RCCL endpoint (dlopen of librccl, ncclCommInitRank, stream / event plumbing)
    and the sharded drivers on it, world size 1, against the oracle."""
    torch = _torch()
    from oracle import zernike_oracle as zo
    from mtflearn_amd import _native, distributed as D
    if rendezvous == "file":
        comm = D.RcclComm(0, 0, 1, path=str(tmp_path / "id"))
        assert not os.path.exists(tmp_path / "id")                           # rank 0 removes it after the join
    elif rendezvous == "tcp":
        comm = D.RcclComm(0, 0, 1, port=_free_port())
    else:
        comm = D.RcclComm(0, 0, 1, unique_id=_native.Comm.unique_id())
    try:
        assert (comm.rank, comm.world) == (0, 1)
        assert comm.allgather_host(b"abcd") == [b"abcd"]
        big = np.random.default_rng(0).integers(0, 255, 3_000_001, dtype=np.uint8).tobytes()     # grows the staging buffer
        assert comm.allgather_host(big) == [big]
        assert comm.allgather_host(b"xy") == [b"xy"] and comm.allgather_host(big[:70000]) == [big[:70000]]
        assert comm.max_over_ranks(2.5) == 2.5
        comm.barrier()
        z = _zps(8, 32)
        plan = z._device_plan()
        rng = np.random.default_rng(3)
        p = rng.random((333, 32, 32), dtype=np.float32)
        got = D.sharded_patch_moments(plan, comm, torch.from_numpy(p).cuda(), 333, n_chunks=3)
        torch.cuda.synchronize()
        rel_close(got.cpu().numpy(), zo.moments_patches(p, z.polynomials))
        img = rng.random((45, 77), dtype=np.float32)
        ref = zo.moments_frame_direct(img, z.polynomials)
        gotf = D.sharded_frame_moments(plan, comm, torch.from_numpy(img).cuda(), n_chunks=2)
        torch.cuda.synchronize()
        rel_close(gotf.cpu().numpy(), ref)
        theta = np.linspace(0, 2 * np.pi, 36, endpoint=False)
        rot, ab, mir = D.sharded_frame_maps(plan, comm, torch.from_numpy(img).cuda(), 25, folds=(2, 3), theta=theta, n_chunks=3)
        torch.cuda.synchronize()
        rel_close(rot.cpu().numpy(), zo.rot_maps(ref, z.n, z.m, [2, 3]), rtol=1e-9, atol_scale=1e-11)
        rel_close(mir.cpu().numpy(), zo.mirror_map(ref, z.n, z.m, theta=theta), rtol=1e-9, atol_scale=1e-11)
        frames = rng.random((3, 40, 50), dtype=np.float32)
        gotb = D.sharded_frames_moments(plan, comm, torch.from_numpy(frames).cuda(), 3)
        torch.cuda.synchronize()
        rel_close(gotb.cpu().numpy(), np.stack([zo.moments_frame_direct(f, z.polynomials) for f in frames]))
    finally:
        comm.close()


def test_device_entry_points_check_their_operands():
    torch = _torch()
    from mtflearn_amd import distributed as D
    z = _zps(4, 8)
    plan = z._device_plan()
    with pytest.raises(TypeError, match="float32 or float64"):
        D.patch_moments_device(plan, torch.zeros((4, 8, 8), dtype=torch.float16, device="cuda"))
    with pytest.raises(ValueError, match="must live on the GPU"):
        D.frame_moments_device(plan, torch.zeros((16, 16)))
    with pytest.raises(ValueError, match="contiguous"):
        D.frame_moments_device(plan, torch.zeros((16, 32), device="cuda")[:, ::2])
    assert torch.cuda.current_device() == 0                                  # entry points leave the device alone


@pytest.mark.parametrize("n_max,size", [(8, 32), (12, 64), (8, 24)])
def test_batch_kernels_write_into_odd_rows_of_the_gathered_matrix(n_max, size):
    """A rank's block inside the gathered (N_total, n_poly) matrix starts at row rank * n_local -- an odd row of an odd
    n_poly is only 8-byte aligned.  The fast batch kernels (row-pair and stream) must take it (not fall back to the
    generic kernel) and produce the same bits as into an aligned buffer."""
    torch = _torch()
    from mtflearn_amd.distributed import patch_moments_device
    z = _zps(n_max, size)
    plan = z._device_plan()
    n_poly = len(z.n)
    assert n_poly % 2 == 1
    n = 98304 + 1237                                       # enough patches for ZK_PATH_AUTO to pick the stream kernel at K = 24
    p = torch.rand((n, size, size), device="cuda")
    aligned = patch_moments_device(plan, p)
    big = torch.full((n + 3, n_poly), float("nan"), dtype=torch.float64, device="cuda")
    for row0 in (1, 2, 3):
        big.fill_(float("nan"))
        patch_moments_device(plan, p, out=big[row0:row0 + n])
        assert torch.equal(big[row0:row0 + n], aligned), row0
        assert torch.isnan(big[:row0]).all() and torch.isnan(big[row0 + n:]).all()      # nothing written outside the block


@pytest.mark.parametrize("dt", [np.uint8, np.uint16, np.int16, np.bool_, np.int8])
def test_detector_formats_are_widened_on_the_device(dt):
    """uint8 / uint16 / int16 inputs cross PCIe as they are and are widened to float32 on the device: the moments equal
    those of the float64 promotion NumPy (the reference) would compute on, for every host entry point, chunked or not."""
    from oracle import zernike_oracle as zo
    rng = np.random.default_rng(5)
    hi = {np.uint8: 255, np.uint16: 65535, np.int16: 32767, np.bool_: 2, np.int8: 127}[dt]
    lo = {np.int16: -32768, np.int8: -128}.get(dt, 0)
    z = _zps(8, 32)
    plan = z._device_plan()
    patches = rng.integers(lo, hi, (3000, 32, 32)).astype(dt)
    frame = rng.integers(lo, hi, (90, 131)).astype(dt)
    pts = np.column_stack([rng.integers(0, 131, 500), rng.integers(0, 90, 500)])
    ref_p = zo.moments_patches(patches.astype(np.float64), z.polynomials)
    ref_f = zo.moments_frame_direct(frame.astype(np.float64), z.polynomials)
    for chunk in (0, 1 << 20):
        plan.set_host_chunk(chunk)
        rel_close(z.transform(patches).data, ref_p)
        rel_close(z.transform(frame).data, ref_f)
        np.testing.assert_array_equal(z.transform_at(frame, pts).data, z.transform_at(frame.astype(np.float32), pts).data)
        a, b = z.symmetry_maps(frame), z.symmetry_maps(frame.astype(np.float32))
        for key in ("rot_maps", "abs", "mirror_map"):
            np.testing.assert_array_equal(a[key], b[key])
    plan.set_host_chunk(0)


@pytest.mark.parametrize("rendezvous", ["file", "tcp"])
def test_two_process_rendezvous_reaches_rccl(tmp_path, rendezvous):
    """Two real processes meet through the file / TCP rendezvous of zk_comm_init_*: both receive rank 0's id and enter
    ncclCommInitRank.  On this one-GPU box RCCL then refuses the communicator (both ranks sit on device 0) -- which is
    exactly what shows that the id travelled: a broken rendezvous would time out or fail to connect instead."""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import sys; sys.path.insert(0, sys.argv[1]);\n"
        "from mtflearn_amd import _native\n"
        "rank = int(sys.argv[2]); kw = dict(path=sys.argv[4]) if sys.argv[3] == 'file' else dict(port=int(sys.argv[4]))\n"
        "try:\n"
        "    _native.Comm(0, rank, 2, timeout=60.0, **kw)\n"
        "    print('CONNECTED')\n"
        "except RuntimeError as exc:\n"
        "    print('ERROR', exc)\n")
    target = str(tmp_path / "id") if rendezvous == "file" else str(_free_port())
    procs = [subprocess.Popen([sys.executable, "-c", code, os.path.join(ROOT, "motif-learn_amd"), str(r), rendezvous, target],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for out in outs:
        assert "timed out" not in out and "cannot" not in out, outs
        # two ranks on one device: RCCL rejects the communicator (or, on a multi-GPU box, would accept it)
        assert "CONNECTED" in out or "ncclCommInitRank" in out, outs
    if rendezvous == "file":
        assert not os.path.exists(target)
