"""BASELINE configs [2] and [4] at their full 4096 x 4096 size, through the C ABI on the GPU.

The oracle cannot finish 16.8 M positions, so parity at full size rests on:
  * an independent kernel: the batch kernel on >= 10^5 strided windows equals the dense kernel there,
  * linearity of the dense kernel on the whole frame,
  * >= 300 oracle positions, chosen to include the zero-padded borders and the rows / columns where the
    even-K alignment of reference ``_zps.py:165-178`` decides which pixels a window covers,
  * row bands (the multi-GPU shards), written IN PLACE into the full array through the plane-stride entry
    points, reassemble the one-launch result bit for bit.
Tolerance: elementwise rtol 1e-6 (north_star) with the 1e-12 * max|Z| floor of conftest.rel_close
(1e-11 at n_max 12, the rounding of the Legendre form there)."""
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


def _positions(H, K, rng, n_random=240):
    """(rows, cols): random interior positions plus every combination of border / alignment coordinates."""
    eb = (K - 1) // 2
    ea = K - 1 - eb
    edge = [0, 1, eb - 1, eb, ea, ea + 1, H // 2, H - 1 - ea - 1, H - 1 - ea, H - 1 - eb, H - 2, H - 1]
    rows = list(rng.integers(0, H, n_random))
    cols = list(rng.integers(0, H, n_random))
    for r in edge:
        for c in (0, ea, H // 3, H - 1 - eb, H - 1):
            rows.append(r)
            cols.append(c)
    for c in edge:
        rows.append(int(rng.integers(0, H)))
        cols.append(c)
    return np.array(rows), np.array(cols)


def _oracle_at(zo, frame, z, rows, cols):
    return zo.moments_frame_at(frame, zo.convolution_basis(z.polynomials, z.n), rows, cols)   # the reference's dense path


@pytest.mark.parametrize("n_max,size,step,floor", [(12, 64, 12, 1e-11), (10, 32, 12, 1e-12)])
def test_dense_full_4096(n_max, size, step, floor):
    """configs[2] (64-px, n_max 12 -> 12.2 GB of moments) and the moments of configs[4] (32-px, n_max 10 ->
    8.9 GB) on the whole 4096^2 frame."""
    import torch
    from oracle import zernike_oracle as zo
    from mtflearn_amd.synthetic import honeycomb_frame
    from mtflearn_amd.distributed import frame_moments_device, patch_moments_device, shard_bounds
    H, K = 4096, size
    z = _zps(n_max, K)
    plan = z._device_plan()
    n_poly = len(z.n)
    frame = honeycomb_frame(H, seed=1)
    dev = torch.device("cuda:0")
    f = torch.from_numpy(frame).to(dev)
    zf = frame_moments_device(plan, f)                                    # (n_poly, 4096, 4096)
    torch.cuda.synchronize()
    scale = zf.abs().max().item()

    # independent kernel: batch moments of the strided windows == dense moments at their centres
    win = f.unfold(0, K, step).unfold(1, K, step)
    nr, nc = win.shape[:2]
    assert nr * nc >= 100000
    zp = patch_moments_device(plan, win.reshape(-1, K, K).contiguous())
    ea = K - 1 - (K - 1) // 2
    at = zf[:, ea:ea + (nr - 1) * step + 1:step, ea:ea + (nc - 1) * step + 1:step].permute(1, 2, 0).reshape(-1, n_poly)
    assert (zp - at).abs().max().item() <= floor * scale
    del zp, at, win

    # oracle: interior, zero-padded borders, even-K alignment rows / columns
    rng = np.random.default_rng(11)
    rows, cols = _positions(H, K, rng)
    assert len(rows) >= 300
    got = zf[:, torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)].T.cpu().numpy()
    rel_close(got, _oracle_at(zo, frame, z, rows, cols), atol_scale=floor)

    # the shards of an 8-GPU run, written in place into one (n_poly, H, W) array, give the same bits
    full = torch.full((n_poly, H, H), float("nan"), dtype=torch.float64, device=dev)
    for rank in range(8):
        start, count, _ = shard_bounds(H, rank, 8)
        frame_moments_device(plan, f, row0=start, n_rows=count, full=full)
    assert torch.equal(full, zf)
    del full

    # linearity on the whole frame (inputs chosen so that 2 g1 + 3 g2 is exact in float32)
    g1 = torch.randint(0, 64, (H, H), device=dev).float() / 256
    g2 = torch.randint(0, 64, (H, H), device=dev).float() / 256
    frame_moments_device(plan, 2 * g1 + 3 * g2, out=zf)
    rhs = frame_moments_device(plan, g1)
    rhs *= 2
    tmp = frame_moments_device(plan, g2)
    rhs.add_(tmp, alpha=3)
    assert (zf - rhs).abs().max().item() <= floor * zf.abs().max().item()


def test_symmetry_maps_full_4096():
    """configs[4] end to end: the fused maps of the whole 4096^2 frame (41 planes, 5.5 GB) against the
    reference's tail (oracle rot_maps / to_complex / mirror_map) at >= 300 positions, and the row-band shards
    written in place."""
    import torch
    from oracle import zernike_oracle as zo
    from mtflearn_amd.synthetic import honeycomb_frame
    from mtflearn_amd.distributed import frame_maps_device, shard_bounds
    H, K, n_max = 4096, 32, 10
    z = _zps(n_max, K)
    plan = z._device_plan()
    frame = honeycomb_frame(H, seed=1)
    dev = torch.device("cuda:0")
    f = torch.from_numpy(frame).to(dev)
    theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
    n_c = sum(k // 2 + 1 for k in range(n_max + 1))
    rot, ab, mir = frame_maps_device(plan, f, n_c, folds=(2, 3, 4, 6), theta=theta)
    rng = np.random.default_rng(12)
    rows, cols = _positions(H, K, rng)
    ref = _oracle_at(zo, frame, z, rows, cols)
    ri, ci = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
    rel_close(rot[:, ri, ci].T.cpu().numpy(), zo.rot_maps(ref, z.n, z.m, [2, 3, 4, 6]), rtol=1e-8, atol_scale=1e-11)
    rel_close(ab[:, ri, ci].T.cpu().numpy(), np.abs(zo.to_complex(ref, z.n, z.m)[0]), rtol=1e-8, atol_scale=1e-11)
    rel_close(mir[ri, ci].cpu().numpy(), zo.mirror_map(ref, z.n, z.m), rtol=1e-8, atol_scale=1e-11)
    nan = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device=dev)
    full = (nan(4, H, H), nan(n_c, H, H), nan(H, H))
    for rank in range(8):
        start, count, _ = shard_bounds(H, rank, 8)
        frame_maps_device(plan, f, n_c, folds=(2, 3, 4, 6), theta=theta, row0=start, n_rows=count, full=full)
    assert torch.equal(full[0], rot) and torch.equal(full[1], ab) and torch.equal(full[2], mir)


def test_clustering_of_the_full_moment_matrix():
    """BASELINE configs[1] carried to its consumer: all 4 068 289 dense 32-px windows of a 2048^2 frame -> moments (device
    resident) -> k-means.  Against scikit-learn's KMeans on the same matrix copied to the host (what the reference's
    ``kmeans_lbs`` computes, ``clustering/_clustering_functions.py:8-22``): tens of Lloyd iterations at the full size."""
    import torch
    from sklearn.cluster import KMeans
    from mtflearn_amd import distributed as D
    from mtflearn_amd.clustering import DeviceRows, kmeans_fit, _relabel_by_size
    from mtflearn_amd.synthetic import honeycomb_frame
    frame = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
    patches = frame.unfold(0, 32, 1).unfold(1, 32, 1).reshape(-1, 32, 32).contiguous()
    plan = _zps(8, 32)._device_plan()
    moments = D.patch_moments_device(plan, patches)
    torch.cuda.synchronize()
    del patches
    assert moments.shape == (4068289, 45)
    with DeviceRows.adopt(moments.data_ptr(), moments.shape[0], moments.shape[1], device=0) as rows:
        labels, centers, n_iter = kmeans_fit(rows, 4, random_state=0)
    ref = KMeans(n_clusters=4, random_state=0).fit(moments.cpu().numpy())
    assert n_iter == ref.n_iter_ and n_iter > 5
    assert np.mean(labels == ref.labels_) >= 0.999999                       # ties at the last bit: at most a handful of 4 M rows
    np.testing.assert_allclose(centers, ref.cluster_centers_, rtol=0, atol=1e-9 * np.abs(ref.cluster_centers_).max())
    sizes = np.bincount(_relabel_by_size(labels))
    assert (np.diff(sizes) <= 0).all() and sizes.sum() == 4068289


def test_symmetry_scores_of_the_full_moment_matrix():
    """The rank-2 symmetry tail at BASELINE size: all 4 068 289 moment vectors of a 2048^2 frame through zk_moment_maps in
    host chunks; 600 random rows against the oracle's rot_maps / to_complex / mirror_map, every row finite, and the key-point
    route (zk_points_maps on the same frame) equal to it at the same positions."""
    from oracle import zernike_oracle as zo
    from mtflearn_amd.synthetic import honeycomb_frame
    z = _zps(8, 32)
    frame = honeycomb_frame(2048, seed=0)
    zm = z.transform(frame)                                              # (45, 2048, 2048), dense
    H = 2048 - 31
    rows = np.ascontiguousarray(zm.data[:, 16:16 + H, 16:16 + H].reshape(45, -1).T)     # (4 068 289, 45): un-padded windows
    assert rows.shape == (4068289, 45)
    got = z.symmetry_of(rows)
    assert got["rot_maps"].shape == (4068289, 4) and got["abs"].shape == (4068289, 25) and got["mirror_map"].shape == (4068289,)
    assert np.isfinite(got["rot_maps"]).all() and np.isfinite(got["mirror_map"]).all()
    rng = np.random.default_rng(5)
    pick = rng.integers(0, len(rows), 600)
    sub = rows[pick]
    rel_close(got["rot_maps"][pick], zo.rot_maps(sub, z.n, z.m, [2, 3, 4, 6], p=2, m_unselect=(0, 1)), rtol=1e-9)
    rel_close(got["abs"][pick], np.abs(zo.to_complex(sub, z.n, z.m)[0]), rtol=1e-12)
    rel_close(got["mirror_map"][pick], zo.mirror_map(sub, z.n, z.m, theta=None, p=2, m_unselect=(0, 1)), rtol=1e-9)
    ii, kk = np.divmod(pick, H)
    at = z.symmetry_at(frame, np.column_stack([kk + 16, ii + 16]))       # (x, y) of the same windows
    rel_close(at["rot_maps"], got["rot_maps"][pick], rtol=1e-6, atol_scale=1e-9)
    rel_close(at["mirror_map"], got["mirror_map"][pick], rtol=1e-6, atol_scale=1e-9)


def test_one_ranks_share_of_the_frame_batch():
    """configs[3] at one rank's FULL share: 8 frames of 2048^2 -> (8, 45, 2048, 2048) float64 (12.1 GB) through the product's
    driver `sharded_frames_moments` on the library's RCCL communicator (world 1 here; worlds 2-4: tests/test_gpu_multirank.py).
    Every frame must equal the dense kernel's own result bit for bit (the driver writes in place into the gathered array), and
    oracle positions of the first and the last frame -- zero-padded borders included -- must match."""
    import torch
    from oracle import zernike_oracle as zo
    from mtflearn_amd import _native, distributed as D
    from mtflearn_amd.synthetic import honeycomb_frame
    H, K, n_max, per_rank = 2048, 32, 8, 8
    z = _zps(n_max, K)
    plan = z._device_plan()
    dev = torch.device("cuda:0")
    host = [honeycomb_frame(H, seed=1000 + i) for i in range(per_rank)]
    frames = torch.stack([torch.from_numpy(f) for f in host]).to(dev)
    comm = D.RcclComm(0, 0, 1, unique_id=_native.Comm.unique_id())
    try:
        full = torch.full((per_rank, len(z.n), H, H), float("nan"), dtype=torch.float64, device=dev)
        D.sharded_frames_moments(plan, comm, frames, per_rank, out=full)
        torch.cuda.synchronize()
    finally:
        comm.close()
    assert torch.isfinite(full).all()
    one = torch.empty((len(z.n), H, H), dtype=torch.float64, device=dev)
    for i in range(per_rank):
        D.frame_moments_device(plan, frames[i], out=one)
        assert torch.equal(one, full[i]), i
    rng = np.random.default_rng(31)
    rows, cols = _positions(H, K, rng)
    ri, ci = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
    for i in (0, per_rank - 1):
        rel_close(full[i][:, ri, ci].T.cpu().numpy(), _oracle_at(zo, host[i], z, rows, cols))


def test_direct_batch_beyond_one_round_of_launches():
    """zk_patch_direct_kernel takes 2^20 patches per launch (all chunks of the set in it): a batch of 2^20 + 300 needs a second
    round whose patch and result offsets must line up.  Rows on both sides of the seam equal the same patches transformed on
    their own (bit for bit: same sums in the same order) and the oracle's plain sum."""
    import torch
    from oracle import zernike_oracle as zo
    from mtflearn_amd import _native as native, distributed as D
    z = _zps(13, 20)                                      # 105 functions = 4 + 3 blocks of 16: both bodies of the kernel
    plan = z._device_plan()
    plan.set_path(native.PATH_DIRECT)
    try:
        n = (1 << 20) + 300
        g = torch.Generator(device="cuda").manual_seed(1)
        p = torch.rand((n, 20, 20), device="cuda", generator=g, dtype=torch.float32)
        out = D.patch_moments_device(plan, p)
        idx = torch.from_numpy(np.r_[0:200, (1 << 20) - 200:(1 << 20) + 300]).cuda()
        sub = p[idx].contiguous()
        alone = D.patch_moments_device(plan, sub)
        torch.cuda.synchronize()
        got = out[idx].cpu().numpy()
        np.testing.assert_array_equal(got, alone.cpu().numpy())
        rel_close(got, zo.moments_patches(sub.cpu().numpy(), z.polynomials))
    finally:
        plan.set_path(native.PATH_AUTO)
