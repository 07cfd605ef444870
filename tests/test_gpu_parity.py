"""Parity of the HIP path (through the C ABI) with the reference: golden vectors captured from the
reference, the CPU oracle on seeded inputs, and size-independent properties at BASELINE sizes.

Tolerance (stated once): elementwise rtol = 1e-6 (north_star), with an absolute floor of
1e-12 * max|Z| that only matters for moments that cancel to ~0.  Typical observed error is ~1e-15.
"""
import warnings

import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from mtflearn_amd import _native
    assert _native.device_count() > 0, "GPU tests need a HIP device"
    return _native


@pytest.fixture(scope="module")
def zo():
    from oracle import zernike_oracle
    return zernike_oracle


def _zps(n_max, size):
    from mtflearn_amd import ZPs
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ZPs(n_max, size)


def _dense_basis(zo, z):
    """What the reference's dense path multiplies a window with (_zps.py:165-178: convolution, then (-1)^n): the
    point-flipped, signed basis -- equal to z.polynomials up to the rounding asymmetry of the reference's own basis
    (1e-13 of max|V| at n_max 10, 1e-8 at 24, 8e-4 at 36), which is what the kernels that sum the caller's numbers restate."""
    return zo.convolution_basis(z.polynomials, z.n)


def _floor(name, n_max):
    """Absolute floor of the criterion as a fraction of max|Z|.  What ZK_PATH_AUTO runs, and every kernel that sums the
    caller's own numbers, is held to 1e-12 up to n_max 12 and 1e-11 above; the polynomial kernels
    forced above n_max 16 -- an explicit opt-in there -- to their documented distance from the reference."""
    if name in ("separable", "stream") and n_max > 16:
        return 3e-7 if n_max > 20 else 1e-8
    return 1e-11 if n_max > 12 else 1e-12


def _both_paths(native, z, array, mode):
    """Moments from every kernel family the plan has for this input (generic / folded / separable),
    keyed by name."""
    plan = z._device_plan()
    run = plan.transform_patches if mode == 0 else plan.transform_frame
    out = {}
    for path, name in native.PATH_NAMES.items():
        if plan.has_path(mode, native.dtype_code(array.dtype), path):
            plan.set_path(path)
            out[name] = run(array)
    plan.set_path(native.PATH_AUTO)
    assert "generic" in out
    out["auto"] = run(array)
    return out


# ------------------------------------------------------------------ golden vectors: batch path
@pytest.mark.parametrize("key_in,key_out,n_max,size,expect_fast", [
    ("blobs_32", "Z_blobs_8_32", 8, 32, True),
    ("rand_f32_70_32", "Z_rand_f32_8_32", 8, 32, True),
    ("rand_f32_70_32", "Z_rand_f32_10_32", 10, 32, True),
    ("rand_f64_5_9", "Z_rand_f64_5_9", 5, 9, True),
    ("rand_f32_6_11", "Z_rand_f32_10_11", 10, 11, False),
    ("rand_f32_3_64", "Z_rand_f32_12_64", 12, 64, True),
])
def test_patches_golden(native, golden, key_in, key_out, n_max, size, expect_fast):
    z = _zps(n_max, size)
    res = _both_paths(native, z, np.ascontiguousarray(golden[key_in]), 0)
    assert ("separable" in res) == expect_fast
    for name, got in res.items():
        assert got.dtype == np.float64 and got.shape == golden[key_out].shape
        rel_close(got, golden[key_out], atol_scale=1e-12)
        # SURVEY 8c's criterion verbatim -- elementwise rtol = 1e-6 with NO absolute floor -- holds too on the
        # reference's outputs (observed worst element: 1e-10)
        np.testing.assert_allclose(got, golden[key_out], rtol=1e-6, atol=0, err_msg=name)


def test_patches_api_dtypes_and_layouts(golden):
    z8 = _zps(8, 32)
    zm = z8.transform(golden["blobs_32"])
    assert type(zm).__name__ == "zmoments" and zm.patch_size == 32 and zm.valid_mask is None
    np.testing.assert_array_equal(zm.n, golden["n_8"])
    np.testing.assert_array_equal(zm.m, golden["m_8"])
    np.testing.assert_allclose(zm.data[0, :5], [0.2939061752406, 0.01330401251592, 0.01135895180554,
                                                0.00551359956386, -0.1162804839817], rtol=1e-10)
    rel_close(zm.rot_maps([2, 3, 4, 6]), golden["pp2_rot_maps"], rtol=1e-9)
    rel_close(zm.mirror_map(), golden["pp2_mirror"], rtol=1e-9)
    # uint8 input (exact in float64), non-contiguous view, float64 copy of a float32 batch
    rel_close(_zps(4, 8).transform(golden["rand_u8_3_8"]).data, golden["Z_rand_u8_4_8"])
    p = golden["rand_f32_70_32"]
    rel_close(z8.transform(p[::-1]).data, golden["Z_rand_f32_8_32"][::-1])
    rel_close(z8.transform(p.astype(np.float64)).data, golden["Z_rand_f32_8_32"])
    rel_close(z8.fit_transform(p).data, golden["Z_rand_f32_8_32"])
    # detector formats: uint16 / int16 / bool go through float32 exactly -> the same moments as their float64 copies
    rng = np.random.default_rng(5)
    u16 = rng.integers(0, 65536, size=(70, 32, 32), dtype=np.uint16)
    np.testing.assert_array_equal(z8.transform(u16).data, z8.transform(u16.astype(np.float32)).data)
    rel_close(z8.transform(u16).data, z8.transform(u16.astype(np.float64)).data)
    frame16 = rng.integers(-3000, 3000, size=(40, 50)).astype(np.int16)
    rel_close(z8.transform(frame16).data, z8.transform(frame16.astype(np.float64)).data)


# ------------------------------------------------------------------ golden vectors: dense path
def test_frame_golden(native, golden, zo):
    z8 = _zps(8, 32)
    img32 = np.ascontiguousarray(golden["frame_f32_48_56"])
    ref = golden["Zf_frame_f64cast_8_32"]
    for arr in (img32, img32.astype(np.float64)):          # float32 values are exact in float64
        res = _both_paths(native, z8, arr, 1)
        assert "folded" in res and "separable" in res
        for got in res.values():
            assert got.shape == ref.shape and got.dtype == np.float64
            rel_close(got, ref)
    # the reference's own float32-image output is only good to ~1e-7 * max|Z| (single-precision FFT)
    zm = z8.transform(img32)
    assert np.abs(zm.data - ref).max() <= 1e-6 * golden["Zf_frame_f32_8_32_maxabs"]
    assert zm.valid_mask.shape == (48, 56) and zm.patch_size == 32
    small = np.ascontiguousarray(golden["frame_f64_20_23"])
    for key, (n_max, size) in {"Zf_small_5_9": (5, 9), "Zf_small_4_8": (4, 8), "Zf_small_10_11": (10, 11)}.items():
        for got in _both_paths(native, _zps(n_max, size), small, 1).values():
            rel_close(got, golden[key])


def test_frame_tail_pipeline_golden(golden):
    z = _zps(6, 12)
    zm = z.transform(golden["frame_f32_24_28"])
    rel_close(zm.data, golden["pp3_moments"])
    np.testing.assert_array_equal(zm.valid_mask, golden["pp3_valid_mask"])
    rel_close(zm.rot_maps([2, 3, 4, 6]), golden["pp3_rot_maps"], rtol=1e-8)
    rel_close(zm.mirror_map(), golden["pp3_mirror"], rtol=1e-8)
    rel_close(np.abs(zm.to_complex().data), np.abs(golden["pp3_complex"]), rtol=1e-8)


def test_moments_at_key_points(native, zo):
    """transform_at == cut the reference's key-point patches (features/_keypoint.py:60-78) + batch path."""
    rng = np.random.default_rng(5)
    for n_max, size, dtype in [(8, 32, np.float32), (10, 32, np.float64), (6, 33, np.float32), (5, 16, np.float32),
                               (12, 64, np.float32), (13, 40, np.float32), (16, 32, np.float32), (17, 40, np.float32),
                               (26, 56, np.float32)]:
        z = _zps(n_max, size)
        frame = rng.random((150, 210)).astype(dtype)
        margin = size // 2 + 2
        pts = np.column_stack([rng.uniform(margin, 210 - margin, 500), rng.uniform(margin, 150 - margin, 500)])
        ipts = np.rint(pts).astype(int)
        s1, s2 = size // 2, size - size // 2
        patches = np.array([frame[y - s1:y + s2, x - s1:x + s2] for x, y in ipts])     # reference slicing
        ref = zo.moments_patches(patches, z.polynomials)
        got = z.transform_at(frame, pts)
        assert got.data.shape == ref.shape and got.patch_size == size
        rel_close(got.data, ref, atol_scale=_floor("auto", n_max))
    z = _zps(8, 32)
    frame = rng.random((64, 64)).astype(np.float32)
    edge = z.transform_at(frame, [[0, 0], [63, 63], [5, 60]]).data                       # zero padding outside
    padded = np.pad(frame, 32)
    refp = np.array([padded[y + 16:y + 48, x + 16:x + 48] for x, y in [(0, 0), (63, 63), (5, 60)]])
    rel_close(edge, zo.moments_patches(refp, z.polynomials))
    # plans without the key-point kernel (n_max > 16) cut the windows on the device: same zero padding
    z18 = _zps(18, 24)
    assert not z18._device_plan().supports(native.OP_POINTS, native.ZK_F32)
    edge18 = z18.transform_at(frame, [[0, 0], [63, 63], [5, 60], [30, 31]]).data
    padded = np.pad(frame, 24)
    ref18 = np.array([padded[y + 12:y + 36, x + 12:x + 36] for x, y in [(0, 0), (63, 63), (5, 60), (30, 31)]])
    rel_close(edge18, zo.moments_patches(ref18, z18.polynomials), atol_scale=1e-11)
    assert z.transform_at(frame, np.empty((0, 2))).data.shape == (0, 45)


@pytest.mark.parametrize("n_max,size,n_points,order", [(8, 32, 100000, "random"), (8, 32, 70001, "sorted"), (12, 40, 30000, "random"),
                                                       (14, 32, 9000, "clustered"), (6, 16, 4096, "random"), (8, 32, 4095, "random")])
def test_key_points_in_bucket_order_are_bit_identical(native, zo, n_max, size, n_points, order, monkeypatch):
    """Large point lists go through a counting sort into buckets of one frame row x 256 columns (zk_sep_points.hip) and the
    moment kernel takes them in bucket order, writing every result to the point's own row: the SAME bits as the kernel in the
    caller's order (ZK_POINTS_NO_BUCKET=1), whatever the order -- random, sorted, heavy duplicates, points outside the frame --
    and a sample against the oracle on the reference's slicing (features/_keypoint.py:60-78)."""
    import torch
    rng = np.random.default_rng(n_points)
    H, W = 300, 1100                                                                 # five 256-column buckets per row
    frame = rng.random((H, W)).astype(np.float32)
    pts = np.column_stack([rng.integers(-20, W + 20, n_points), rng.integers(-20, H + 20, n_points)]).astype(np.int32)
    if order == "sorted":
        pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    if order == "clustered":
        pts[:, 0] = 500 + pts[:, 0] % 7                                              # thousands of points in two buckets, duplicates
        pts[:, 1] = 100 + pts[:, 1] % 3
    z = _zps(n_max, size)
    plan = z._device_plan()
    d_img, d_pts = torch.from_numpy(frame).cuda(), torch.from_numpy(pts).cuda()
    out = [torch.full((n_points, len(z.n)), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
    from ctypes import c_void_p
    # default (bucket order, whole window rows as 16-byte loads) | caller's order | caller's order, 4-byte loads (round 1's kernel)
    for k, env in enumerate(({}, {"ZK_POINTS_NO_BUCKET": "1"}, {"ZK_POINTS_NO_BUCKET": "1", "ZK_POINTS_NO_WIDE": "1"})):
        for name in ("ZK_POINTS_NO_BUCKET", "ZK_POINTS_NO_WIDE"):
            monkeypatch.delenv(name, raising=False)
        for name, val in env.items():
            monkeypatch.setenv(name, val)
        native.check(plan._lib.zk_transform_points_dev(plan._h, c_void_p(d_img.data_ptr()), native.ZK_F32, H, W, c_void_p(d_pts.data_ptr()),
                                                       n_points, c_void_p(out[k].data_ptr()), None), "zk_transform_points_dev")
        torch.cuda.synchronize()
    for name in ("ZK_POINTS_NO_BUCKET", "ZK_POINTS_NO_WIDE"):
        monkeypatch.delenv(name, raising=False)
    a, b, c = (o.cpu().numpy() for o in out)
    assert not np.isnan(a).any()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    pick = rng.choice(n_points, 200, replace=False)
    s1, s2 = size // 2, size - size // 2
    padded = np.pad(frame, size + 20)
    o = size + 20
    win = np.array([padded[y + o - s1:y + o + s2, x + o - s1:x + o + s2] for x, y in pts[pick]])
    rel_close(a[pick], zo.moments_patches(win, z.polynomials), atol_scale=_floor("auto", n_max))


def test_symmetry_of_moment_rows(native, golden, zo):
    """zk_moment_maps / zk_points_maps: the zmoments tail on rank-2 data (reference _zmoments.py:300-316, 420-493) --
    golden vectors of the reference (its rot_maps / mirror_map / to_complex on the moments of four blob patches), then the
    oracle's tail on oracle moments for every n_max group, every output subset, ragged counts and several host chunks."""
    z = _zps(8, 32)
    mom = golden["Z_blobs_8_32"]
    got = z.symmetry_of(mom)
    rel_close(got["rot_maps"], golden["pp2_rot_maps"], rtol=1e-8)
    rel_close(got["mirror_map"], golden["pp2_mirror"], rtol=1e-8)
    rel_close(got["abs"], np.abs(golden["pp2_complex"]), rtol=1e-8)
    np.testing.assert_array_equal(got["abs_n"], golden["pp2_complex_n"])
    np.testing.assert_array_equal(got["abs_m"], golden["pp2_complex_m"])
    rel_close(z.symmetry_of(mom, n_folds=[3, 6], p=None, mirror=False)["rot_maps"], golden["pp2_rot_maps_pnone"], rtol=1e-8)
    rel_close(z.symmetry_of(mom, n_folds=[3], m_unselect=(0, 1, 2), mirror=False)["rot_maps"], golden["pp2_rot_maps_unsel012"],
              rtol=1e-8)
    # the whole flow from the patches, moments never on the host: frame with the four blobs pasted in, points at their centres
    frame = np.zeros((80, 200), dtype=np.float32)
    pts = []
    for i, blob in enumerate(golden["blobs_32"]):
        frame[20:52, 10 + 45 * i:42 + 45 * i] = blob
        pts.append((10 + 45 * i + 16, 20 + 16))
    at = z.symmetry_at(frame, pts)
    rel_close(at["rot_maps"], golden["pp2_rot_maps"], rtol=1e-8)
    rel_close(at["mirror_map"], golden["pp2_mirror"], rtol=1e-8)
    rel_close(at["abs"], np.abs(golden["pp2_complex"]), rtol=1e-8)

    rng = np.random.default_rng(314)
    for n_max, size, n_rows in [(4, 16, 1), (6, 12, 257), (8, 32, 3000), (10, 32, 700), (12, 40, 513), (14, 32, 300),
                                (16, 36, 255), (20, 44, 130), (23, 48, 70),
                                # above 24 (rows kernels 28 .. 40; |m| beyond 31 in m_unselect: the 64-bit mask)
                                (25, 50, 65), (28, 56, 130), (31, 62, 64), (36, 72, 100), (40, 80, 67)]:
        zz = _zps(n_max, size)
        if n_max > 24:
            assert zz._device_plan().supports(native.OP_MAPS, native.ZK_F64)
        mom = rng.standard_normal((n_rows, len(zz.n))) * np.exp(rng.uniform(-3, 3, (n_rows, 1)))
        for kw in (dict(), dict(n_folds=[2, 5], m_unselect=(0, 2), p=None, theta=np.linspace(0, 2 * np.pi, 48, endpoint=False)),
                   dict(n_folds=[1, 2, 3, 4, 5, 6, 7, 8], theta=np.linspace(0, np.pi, 37)),
                   dict(n_folds=[3, 4], m_unselect=(0, 1, 24, 31, 33, 40))):
            got = zz.symmetry_of(mom, **kw)
            folds, unsel, pp = kw.get("n_folds", [2, 3, 4, 6]), kw.get("m_unselect", (0, 1)), kw.get("p", 2)
            rel_close(got["rot_maps"], zo.rot_maps(mom, zz.n, zz.m, list(folds), p=pp, m_unselect=unsel), rtol=1e-9)
            rel_close(got["abs"], np.abs(zo.to_complex(mom, zz.n, zz.m)[0]), rtol=1e-12)
            rel_close(got["mirror_map"], zo.mirror_map(mom, zz.n, zz.m, theta=kw.get("theta"), p=pp, m_unselect=unsel), rtol=1e-9)
        only = zz.symmetry_of(mom, n_folds=None, abs_moments=False)
        assert set(only) == {"mirror_map"} and only["mirror_map"].shape == (n_rows,)
        only = zz.symmetry_of(mom, mirror=False, abs_moments=False, n_folds=[4])
        assert set(only) == {"rot_maps"} and only["rot_maps"].shape == (n_rows, 1)
    # several host chunks == one chunk, and a zmoments operand; a NaN row stays in its row
    zz = _zps(8, 32)
    mom = rng.standard_normal((5000, 45))
    mom[17, 3] = np.nan
    one = zz.symmetry_of(mom)
    plan = zz._device_plan()
    plan.set_host_chunk(1 << 18)
    try:
        many = zz.symmetry_of(zz.transform(np.zeros((1, 32, 32), dtype=np.float32))._like(mom))
    finally:
        plan.set_host_chunk(0)
    for key in ("rot_maps", "abs", "mirror_map"):
        np.testing.assert_array_equal(one[key], many[key])
    assert np.isnan(one["rot_maps"][17]).all() and np.isnan(one["mirror_map"][17]) and np.isfinite(np.delete(one["rot_maps"], 17, 0)).all()
    assert zz.symmetry_of(np.empty((0, 45)))["rot_maps"].shape == (0, 4)
    with pytest.raises(ValueError):
        zz.symmetry_of(mom[:, :44])
    with pytest.raises(ValueError):
        zz.symmetry_of(mom, m_unselect=(1,))

    # key points end to end against the oracle (reference slicing -> oracle moments -> oracle tail), incl. plans that cut
    # the windows on the device (n_max > 16) and a chunked run
    for n_max, size, dtype in [(8, 32, np.float32), (12, 40, np.float64), (16, 32, np.float32), (18, 40, np.float32)]:
        zz = _zps(n_max, size)
        frame = (rng.random((150, 210)) + 0.1).astype(dtype)
        margin = size // 2 + 2
        ipts = np.column_stack([rng.integers(margin, 210 - margin, 1200), rng.integers(margin, 150 - margin, 1200)])
        s1, s2 = size // 2, size - size // 2
        patches = np.array([frame[y - s1:y + s2, x - s1:x + s2] for x, y in ipts])
        o_mom = zo.moments_patches(patches, zz.polynomials)
        floor = 1e-7 if n_max > 16 else 1e-9 if n_max > 12 else 1e-10
        plan = zz._device_plan()
        for chunk in (0, 1 << 18):
            plan.set_host_chunk(chunk)
            try:
                got = zz.symmetry_at(frame, ipts)
            finally:
                plan.set_host_chunk(0)
            rel_close(got["rot_maps"], zo.rot_maps(o_mom, zz.n, zz.m, [2, 3, 4, 6], p=2, m_unselect=(0, 1)), atol_scale=floor)
            rel_close(got["abs"], np.abs(zo.to_complex(o_mom, zz.n, zz.m)[0]), atol_scale=floor)
            rel_close(got["mirror_map"], zo.mirror_map(o_mom, zz.n, zz.m, theta=None, p=2, m_unselect=(0, 1)), atol_scale=floor)
    assert zz.symmetry_at(frame, np.empty((0, 2)))["rot_maps"].shape == (0, 4)


def test_strided_grid_matches_reference_extractor(native, zo):
    """transform_grid == reference extract_patches(image, size, step) (denoise/_denoise_svd.py:15-49,
    restated here) followed by the batch transform."""
    rng = np.random.default_rng(8)
    for size, n_max, shape, step in [(32, 8, (100, 131), 7), (16, 6, (64, 64), 16), (33, 8, (70, 90), 5),
                                     (32, 10, (32, 40), 3)]:
        z = _zps(n_max, size)
        frame = rng.random(shape).astype(np.float32)

        def starts(extent):
            last = extent - size
            idx = np.arange(0, last, step)
            return idx if idx.size and idx[-1] == last else np.append(idx, last)

        ii, jj = starts(shape[0]), starts(shape[1])
        patches = np.array([frame[i:i + size, j:j + size] for i in ii for j in jj])
        got = z.transform_grid(frame, step)
        assert got.data.shape == (len(ii) * len(jj), len(z.n))
        rel_close(got.data, zo.moments_patches(patches, z.polynomials))


def test_fused_symmetry_maps(native, golden, zo):
    """Fused frame -> maps kernel against the reference's tail on the golden frame; on random frames against the oracle
    (oracle moments through the oracle's tail) and, for all option combinations, against the host container composed with
    the device transform."""
    z = _zps(6, 12)
    img = golden["frame_f32_24_28"]
    maps = z.symmetry_maps(img)
    rel_close(maps["rot_maps"], golden["pp3_rot_maps"], rtol=1e-8)
    rel_close(maps["mirror_map"], golden["pp3_mirror"], rtol=1e-8)
    rel_close(maps["abs"], np.abs(golden["pp3_complex"]), rtol=1e-8)
    np.testing.assert_array_equal(maps["valid_mask"], golden["pp3_valid_mask"])
    zc_n, zc_m, _ = zo.to_complex(golden["pp3_moments"], z.n, z.m)[1], zo.to_complex(golden["pp3_moments"], z.n, z.m)[2], None
    np.testing.assert_array_equal(maps["abs_n"], zc_n)
    np.testing.assert_array_equal(maps["abs_m"], zc_m)
    rng = np.random.default_rng(77)
    for n_max, size, shape, dtype in [(8, 32, (70, 90), np.float32), (10, 32, (40, 130), np.float64),
                                      (5, 16, (33, 65), np.float32), (7, 33, (40, 50), np.float32),
                                      (12, 40, (50, 70), np.float32), (14, 32, (40, 60), np.float32),
                                      (16, 36, (44, 48), np.float64),
                                      (18, 40, (45, 50), np.float32), (20, 44, (50, 47), np.float64),
                                      (23, 48, (50, 52), np.float32),
                                      # above 24: moments from the matrix-core plain sum + planes kernels 28 .. 40
                                      (26, 52, (54, 70), np.float32), (32, 64, (70, 66), np.float64),
                                      (36, 72, (80, 75), np.float32), (40, 80, (84, 90), np.float32)]:
        zz = _zps(n_max, size)
        if n_max > 24:
            assert zz._device_plan().supports(native.OP_MAPS, native.dtype_code(np.dtype(dtype)))
        frame = (rng.random(shape) + 0.1).astype(dtype)
        zm = zz.transform(frame)
        # independent of the product: the oracle's moments of this frame through the oracle's restatement of the
        # reference tail (_zmoments.py:300-316, 420-493)
        o_mom = zo.moments_frame_direct(frame, _dense_basis(zo, zz))
        # (n_max <= 16: the fused kernel sums the exact polynomial; above: the moments are the plain sum over the caller's basis)
        floor = 1e-9 if n_max > 12 else 1e-10
        for kw in (dict(), dict(n_folds=[2, 5], m_unselect=(0, 2), p=None, theta=np.linspace(0, 2 * np.pi, 48, endpoint=False))):
            got = zz.symmetry_maps(frame, **kw)
            folds, unsel, pp = kw.get("n_folds", [2, 3, 4, 6]), kw.get("m_unselect", (0, 1)), kw.get("p", 2)
            rel_close(got["rot_maps"], zo.rot_maps(o_mom, zz.n, zz.m, list(folds), p=pp, m_unselect=unsel), rtol=1e-6, atol_scale=floor)
            rel_close(got["abs"], np.abs(zo.to_complex(o_mom, zz.n, zz.m)[0]), rtol=1e-6, atol_scale=floor)
            rel_close(got["mirror_map"], zo.mirror_map(o_mom, zz.n, zz.m, theta=kw.get("theta"), p=pp, m_unselect=unsel),
                      rtol=1e-6, atol_scale=floor)
        for kwargs in [dict(), dict(n_folds=[3], p=None), dict(n_folds=[2, 3, 4, 5, 6, 8], m_unselect=(0, 1, 2)),
                       dict(n_folds=None, abs_moments=False), dict(mirror=False, n_folds=[1, 2]),
                       dict(theta=np.linspace(0, np.pi, 37))]:
            got = zz.symmetry_maps(frame, **kwargs)
            folds = kwargs.get("n_folds", (2, 3, 4, 6))
            unsel = kwargs.get("m_unselect", (0, 1))
            pp = kwargs.get("p", 2)
            if folds is not None:
                rel_close(got["rot_maps"], zm.rot_maps(folds, p=pp, m_unselect=unsel), rtol=1e-8, atol_scale=1e-11)
            else:
                assert "rot_maps" not in got
            if kwargs.get("abs_moments", True):
                rel_close(got["abs"], np.abs(zm.to_complex().data), rtol=1e-8, atol_scale=1e-11)
            else:
                assert "abs" not in got
            if kwargs.get("mirror", True):
                rel_close(got["mirror_map"], zm.mirror_map(theta=kwargs.get("theta"), p=pp, m_unselect=unsel),
                          rtol=1e-8, atol_scale=1e-11)
            else:
                assert "mirror_map" not in got
    with pytest.raises(ValueError, match="m=0 must be included"):
        z.symmetry_maps(img, m_unselect=(1,))


@pytest.mark.parametrize("n_max,size,height", [(17, 20, 400), (28, 56, 150)])
def test_symmetry_maps_row_bands_above_n_max_16(native, n_max, size, height):
    """n_max 17-24: the maps come from dense class passes into a scratch matrix (<= 1 GiB, i.e. row bands on a
    wide frame) + the planes kernel; 25-40 the same with the moments from the matrix-core plain sum.  A 4096-wide frame
    needs several bands; a 512-column crop needs one: the maps must agree wherever the windows see the same pixels."""
    z = _zps(n_max, size)
    assert z._device_plan().supports(native.OP_MAPS, native.ZK_F32)
    rng = np.random.default_rng(21)
    frame = (rng.random((height, 4096)) + 0.2).astype(np.float32)
    wide = z.symmetry_maps(frame)
    crop = z.symmetry_maps(np.ascontiguousarray(frame[:, 1000:1512]))
    for key in ("rot_maps", "abs", "mirror_map"):
        a, b = wide[key][..., 1000 + size:1512 - size], crop[key][..., size:512 - size]
        assert a.shape == b.shape
        rel_close(a, b, rtol=1e-10, atol_scale=1e-12)


# ------------------------------------------------------------------ oracle on seeded inputs, edge cases
@pytest.mark.parametrize("n_patches", [1, 2, 63, 64, 65, 130, 257, 1000])
def test_patches_ragged_counts(native, zo, n_patches):
    rng = np.random.default_rng(n_patches)
    z = _zps(8, 32)
    p = rng.random((n_patches, 32, 32), dtype=np.float32) - 0.25
    ref = zo.moments_patches(p, z.polynomials)
    for got in _both_paths(native, z, p, 0).values():
        rel_close(got, ref)


@pytest.mark.parametrize("n_max,size,dtype", [
    (8, 64, np.float32),      # RUN=4 batch kernel (64-B runs)
    (10, 64, np.float32),
    (4, 32, np.float32), (5, 32, np.float32), (6, 32, np.float32), (7, 32, np.float32), (9, 48, np.float32),
    (9, 32, np.float32), (1, 32, np.float32), (3, 64, np.float32), (10, 96, np.float32),
    (8, 16, np.float32), (8, 20, np.float32), (10, 24, np.float32), (6, 28, np.float32), (8, 36, np.float32),
    (10, 40, np.float32), (10, 72, np.float32), (7, 100, np.float32), (4, 128, np.float32),
    (8, 32, np.float64), (10, 64, np.float64), (6, 8, np.float64), (8, 10, np.float64), (5, 18, np.float64),
    (10, 30, np.float64), (9, 72, np.float64), (8, 34, np.float32), (4, 12, np.float32),
    (8, 33, np.float32), (6, 17, np.float32), (10, 21, np.float32), (8, 31, np.float32), (10, 65, np.float32),
    (8, 33, np.float64), (5, 9, np.float64), (7, 15, np.float64), (8, 18, np.float32), (6, 22, np.float32),
    (12, 64, np.float32), (11, 32, np.float32), (12, 33, np.float64),
    (8, 32, np.float64), (3, 16, np.float32), (6, 33, np.float32), (8, 72, np.float32), (0, 5, np.float64),
    (12, 64, np.float64), (0, 1, np.float32),
    # n_max 13-16 (the reference's estimate_n_max returns at least 12, _estimate_n_max.py:95): kernels 14 / 16
    (13, 32, np.float32), (14, 32, np.float32), (16, 32, np.float32), (15, 48, np.float32), (16, 33, np.float32),
    (14, 64, np.float32), (16, 72, np.float64), (16, 40, np.float64),
    # n_max 17-24: one pass per mirror-parity class (kernels 20 / 24); above that the generic kernel
    (17, 36, np.float32), (18, 40, np.float32), (20, 48, np.float32), (22, 48, np.float64), (24, 56, np.float32),
    (19, 39, np.float32), (24, 64, np.float64), (25, 56, np.float32),
    # what the reference's own estimator returns for 56 / 64 / 72-px patches (size / 2, _estimate_n_max.py:95,123).  There the
    # reference's float64 basis is no polynomial any more -- and not even mirror-symmetric (1e-6 of max|V| at n_max 28, 8e-4 at
    # 36: DESIGN.md section 7) -- so only a sum over the caller's own numbers, pixel by pixel, matches it: the generic kernel
    (28, 56, np.float32), (32, 64, np.float32), (36, 72, np.float32), (36, 72, np.float64),
])
def test_patches_shapes_vs_oracle(native, zo, n_max, size, dtype):
    rng = np.random.default_rng(100 * n_max + size)
    z = _zps(n_max, size)
    p = (rng.random((77, size, size)) - 0.3).astype(dtype)
    ref = zo.moments_patches(p, z.polynomials)
    res = _both_paths(native, z, p, 0)
    if n_max > 12:
        assert ("separable" in res) == (n_max <= 24) and ("stream" in res) == (n_max <= 16)
        assert ("direct" in res) == (size >= 16)
        expect = native.PATH_DIRECT if n_max >= 17 else (native.PATH_SEPARABLE, native.PATH_STREAM)
        assert z._device_plan().best_path(0, native.dtype_code(np.dtype(dtype)), 77) in np.atleast_1d(expect)
    for name, got in res.items():
        rel_close(got, ref, atol_scale=_floor(name, n_max))


@pytest.mark.parametrize("n_max,size,shape,dtype", [
    (8, 32, (32, 32), np.float32),        # exactly one window fits
    (8, 32, (33, 100), np.float32), (8, 32, (67, 129), np.float64),
    (10, 32, (40, 70), np.float32), (12, 64, (70, 66), np.float32), (10, 72, (80, 90), np.float32),
    (7, 33, (50, 41), np.float32), (5, 9, (9, 9), np.float64), (11, 24, (30, 200), np.float32),
    (2, 3, (5, 7), np.float32), (0, 1, (3, 4), np.float64),
    (14, 32, (45, 70), np.float32), (16, 32, (64, 64), np.float64), (16, 48, (50, 130), np.float32),
    (13, 33, (40, 40), np.float32),
    (18, 40, (50, 70), np.float32), (20, 44, (44, 60), np.float64), (24, 50, (60, 66), np.float32),
    (21, 45, (50, 50), np.float32), (25, 52, (56, 60), np.float32),
    (28, 56, (60, 70), np.float32), (32, 64, (66, 64), np.float32), (36, 72, (72, 80), np.float64),
])
def test_frame_shapes_vs_oracle(native, zo, n_max, size, shape, dtype):
    rng = np.random.default_rng(size * 1000 + shape[0])
    z = _zps(n_max, size)
    img = (rng.random(shape) - 0.5).astype(dtype)
    ref = zo.moments_frame_direct(img, _dense_basis(zo, z))
    for name, got in _both_paths(native, z, img, 1).items():
        rel_close(got, ref, atol_scale=_floor(name, n_max))


@pytest.mark.parametrize("n_max,size,dtype", [(6, 24, np.float32), (8, 33, np.float32), (10, 12, np.float64)])
def test_large_batch_auto_path_is_stream(native, zo, n_max, size, dtype):
    """Batches that fill the chip (>= 98304 patches) of a size without whole-line row units: ZK_PATH_AUTO
    runs the stream kernel; it must agree with the row-pair / generic kernels and with the oracle, tail
    wave included."""
    z = _zps(n_max, size)
    plan = z._device_plan()
    code = native.dtype_code(np.dtype(dtype))
    assert plan.has_path(0, code, native.PATH_STREAM)
    rng = np.random.default_rng(11)
    n = 98304 + 1237
    patches = rng.random((n, size, size)).astype(dtype)
    auto = plan.transform_patches(patches)
    plan.set_path(native.PATH_STREAM)
    forced = plan.transform_patches(patches)
    plan.set_path(native.PATH_GENERIC)
    generic = plan.transform_patches(patches)
    plan.set_path(native.PATH_AUTO)
    np.testing.assert_array_equal(auto, forced)          # AUTO took the stream kernel: bit-identical
    rel_close(auto, generic)
    pick = np.r_[0:150, n - 150:n]
    rel_close(auto[pick], zo.moments_patches(patches[pick], z.polynomials))


@pytest.mark.parametrize("n_max,size,dtype", [(8, 32, np.float32), (8, 24, np.float32), (6, 33, np.float32),
                                              (8, 64, np.float32), (6, 16, np.float64)])
def test_non_finite_pixels(native, n_max, size, dtype):
    """A NaN outside the unit disk is never used by the fast kernels; a NaN inside
    it reaches every moment of its patch, as in the reference."""
    z = _zps(n_max, size)
    disk = np.any(z.polynomials != 0, axis=0)
    rng = np.random.default_rng(12)
    clean = rng.random((130, size, size)).astype(dtype)
    dirty = clean.copy()
    dirty[:, ~disk] = np.nan
    inside = clean.copy()
    rows, cols = np.nonzero(disk)
    for p in range(inside.shape[0]):
        k = rng.integers(len(rows))
        inside[p, rows[k], cols[k]] = np.nan
    for name, path in [("separable", native.PATH_SEPARABLE), ("stream", native.PATH_STREAM)]:
        plan = z._device_plan()
        if not plan.has_path(0, native.dtype_code(np.dtype(dtype)), path):
            continue
        plan.set_path(path)
        try:
            ref = plan.transform_patches(clean)
            np.testing.assert_array_equal(plan.transform_patches(dirty), ref, err_msg=name)
            assert np.isnan(plan.transform_patches(inside)).all(), name
        finally:
            plan.set_path(native.PATH_AUTO)


def test_strip_and_one_output_dense_kernels_agree(native, zo):
    """n_max <= 8, windows up to 65 px: ZK_PATH_SEPARABLE runs the strip kernel (two outputs per lane);
    ZK_NO_STRIP selects the one-output kernel.  Both against the oracle, odd / even sizes and ragged shapes."""
    import os
    rng = np.random.default_rng(31)
    for n_max, size, shape, dtype in [(8, 32, (70, 131), np.float32), (7, 33, (41, 66), np.float32),
                                      (4, 9, (17, 20), np.float64), (8, 64, (66, 70), np.float32),
                                      (6, 65, (65, 65), np.float64), (2, 8, (9, 300), np.float32)]:
        z = _zps(n_max, size)
        plan = z._device_plan()
        img = (rng.random(shape) - 0.5).astype(dtype)
        ref = zo.moments_frame_direct(img, z.polynomials)
        plan.set_path(native.PATH_SEPARABLE)
        try:
            strip = plan.transform_frame(img)
            os.environ["ZK_NO_STRIP"] = "1"
            single = plan.transform_frame(img)
        finally:
            os.environ.pop("ZK_NO_STRIP", None)
            plan.set_path(native.PATH_AUTO)
        rel_close(strip, ref)
        rel_close(single, ref)
        assert np.abs(strip - single).max() > 0   # two different kernels ran (different summation orders)


@pytest.mark.parametrize("n_max,size,shape,dtype", [(8, 32, (70, 131), np.float32), (8, 32, (33, 64), np.float64),
                                                    (6, 24, (41, 66), np.float32), (4, 16, (17, 200), np.float32),
                                                    (8, 8, (9, 70), np.float64), (7, 30, (64, 64), np.float32)])
def test_strip_kernel_with_the_table_in_vgpr_lanes(native, zo, n_max, size, shape, dtype, monkeypatch):
    """ZK_STRIP_V3=1 (opt-in, zk_frame_strip3_kernel: even windows <= 32 px, n_max <= 8): the x table in VGPR lanes read through
    v_fmac_f64_dpp row_newbcast, one sweep per frame row, LDS requests two blocks ahead.  The same operations in the same
    order as the default strip kernel, so the same BITS; both against the oracle; ragged shapes and zero-padded borders."""
    rng = np.random.default_rng(size * 7 + n_max)
    z = _zps(n_max, size)
    plan = z._device_plan()
    img = (rng.random(shape) - 0.5).astype(dtype)
    ref = zo.moments_frame_direct(img, z.polynomials)
    plan.set_path(native.PATH_SEPARABLE)
    count = native.load().zk_debug_strip3_launches
    try:
        before = count()
        base = plan.transform_frame(img)
        assert count() == before                  # the default kernel
        monkeypatch.setenv("ZK_STRIP_V3", "1")
        lanes = plan.transform_frame(img)
        assert count() > before                   # the opt-in kernel really ran
    finally:
        monkeypatch.delenv("ZK_STRIP_V3", raising=False)
        plan.set_path(native.PATH_AUTO)
    rel_close(base, ref)
    np.testing.assert_array_equal(lanes, base)


def test_zero_and_constant_inputs(native):
    z = _zps(8, 32)
    assert not z.transform(np.zeros((5, 32, 32), np.float32)).data.any()
    assert not z.transform(np.zeros((40, 50), np.float32)).data.any()
    ones = z.transform(np.ones((3, 32, 32), np.float32)).data
    ref = z.polynomials.reshape(45, -1).sum(axis=1) / (np.pi * 32 ** 2 / 4)
    rel_close(ones, np.broadcast_to(ref, (3, 45)))


def test_profile_counters(native):
    z = _zps(8, 32)
    plan = z._device_plan()
    plan.profile(True)
    for _ in range(3):
        plan.transform_patches(np.ones((128, 32, 32), np.float32))
    launches, ms = plan.profile_read()
    plan.profile(False)
    assert launches == 3 and ms > 0.0
    assert plan.profile_read() == (0, 0.0)
    assert plan.disk_pixels == 740


def test_c_abi_argument_errors(native):
    import ctypes
    lib = native.load()
    z = _zps(4, 8)
    plan = z._device_plan()
    buf = np.zeros((2, 8, 8), np.float32)
    out = np.zeros((2, 15))
    assert lib.zk_transform_patches(plan._h, buf.ctypes.data_as(ctypes.c_void_p), 7, 2,
                                    out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == -10001
    assert b"dtype" in lib.zk_last_error_string()
    assert lib.zk_transform_patches(plan._h, None, 0, 2, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == -10001
    assert lib.zk_transform_patches(plan._h, None, 0, 0, None) == 0              # empty batch is a no-op
    handle = ctypes.c_void_p()
    n = np.zeros(15, np.int32)
    assert lib.zk_plan_create(8, 15, n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                              n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                              z.polynomials.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 99,
                              ctypes.byref(handle)) == -10001                     # bad device index
    with pytest.raises(RuntimeError, match="not available"):
        big = _zps(25, 64)._device_plan()                                          # no fast tables above n_max 24
        big.set_path(native.PATH_SEPARABLE)
        big.transform_patches(np.zeros((2, 64, 64), np.float32))


def test_size_sweep_fast_paths_agree_with_generic(native):
    """Every patch size 2..70 (odd and even, both dtypes, several n_max): whatever fast kernels a plan
    offers must agree with the generic kernel (itself checked against the oracle above) -- exercises the
    unit / row tables of the folded and separable paths at every alignment."""
    rng = np.random.default_rng(2026)
    checked = 0
    for size in range(2, 71):
        for n_max in sorted({min(size, v) for v in (0, 3, 8, 11)}):
            z = _zps(n_max, size)
            for dtype in (np.float32, np.float64):
                p = (rng.random((67, size, size)) - 0.4).astype(dtype)
                res = _both_paths(native, z, p, 0)
                img = (rng.random((size + 3, size + 66)) - 0.4).astype(dtype)
                resf = _both_paths(native, z, img, 1)
                for out in (res, resf):
                    ref = out["generic"]
                    scale = np.abs(ref).max()
                    for name, got in out.items():
                        if name != "generic":
                            tol = 1e-12 * scale
                            assert np.abs(got - ref).max() <= tol, (size, n_max, dtype, name)
                            checked += 1
    assert checked > 500


def test_random_plans_every_family_agrees_with_generic(native):
    """Seeded random (size, n_max <= 24, dtype, N, frame shape): every kernel family a plan offers -- row-pair,
    stream, class-pass, folded, one- or two-wave builds of every n_max kernel -- against the generic kernel."""
    rng = np.random.default_rng(77)
    families = set()
    for _ in range(60):
        size = int(rng.integers(8, 81))
        n_max = int(min(size, rng.integers(0, 25)))
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        z = _zps(n_max, size)
        p = (rng.random((int(rng.integers(1, 200)), size, size)) - 0.4).astype(dtype)
        img = (rng.random((int(rng.integers(size, size + 40)), int(rng.integers(size, size + 100)))) - 0.4).astype(dtype)
        for out in (_both_paths(native, z, p, 0), _both_paths(native, z, img, 1)):
            ref = out["generic"]
            for name, got in out.items():
                if name != "generic":
                    assert np.abs(got - ref).max() <= _floor(name, n_max) * np.abs(ref).max(), (size, n_max, dtype, name)
                    families.add(name)
    assert families == {"separable", "stream", "folded", "direct", "auto"}


def test_plain_c_client(native, zo, tmp_path):
    """The C ABI from a C program (gcc, no Python in that process): tests/c_abi_check.c."""
    import os
    import subprocess
    from conftest import ROOT
    z = _zps(8, 32)
    rng = np.random.default_rng(11)
    patches = rng.random((130, 32, 32), dtype=np.float32)
    image = rng.random((50, 70), dtype=np.float32)
    src, dst, exe = tmp_path / "in.bin", tmp_path / "out.bin", tmp_path / "c_abi_check"
    with open(src, "wb") as f:
        np.array([32, 45, 130, 50, 70], np.int32).tofile(f)
        z.n.astype(np.int32).tofile(f)
        z.m.astype(np.int32).tofile(f)
        z.polynomials.tofile(f)
        patches.tofile(f)
        image.tofile(f)
    libdir = os.path.dirname(native.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "c_abi_check.c"),
                           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lzernike_hip",
                           f"-Wl,-rpath,{libdir}"])
    subprocess.check_call([str(exe), str(src), str(dst)])
    out = np.fromfile(dst)
    rel_close(out[:130 * 45].reshape(130, 45), zo.moments_patches(patches, z.polynomials))
    rel_close(out[130 * 45:].reshape(45, 50, 70), zo.moments_frame_direct(image, z.polynomials))


# ------------------------------------------------------------------ BASELINE sizes: properties on device
def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_full_size_batch_equals_dense_and_linearity(native, zo):
    """Config 2 (2048 x 2048 frame, 32-px, n_max=8): the batch kernel on all 4 068 289 sliding
    windows and the dense kernel on the frame are independent kernels computing the same numbers at
    the un-padded positions; both are linear in the image; 300 random positions match the oracle."""
    torch = _torch()
    from mtflearn_amd.synthetic import honeycomb_frame
    from mtflearn_amd.distributed import patch_moments_device, frame_moments_device
    K, H = 32, 2048
    z = _zps(8, K)
    plan = z._device_plan()
    frame = honeycomb_frame(H, seed=0)
    dev = torch.device("cuda:0")
    f = torch.from_numpy(frame).to(dev)
    win = f.unfold(0, K, 1).unfold(1, K, 1)                      # (H-K+1, H-K+1, K, K) view
    nwin = H - K + 1
    patches = win.reshape(-1, K, K).contiguous()                 # 16.7 GB
    assert patches.shape[0] == nwin * nwin == 4068289
    zp = patch_moments_device(plan, patches)                     # (N, 45)
    zf = frame_moments_device(plan, f)                           # (45, H, W)
    torch.cuda.synchronize()
    ea = K - 1 - (K - 1) // 2
    inner = zf[:, ea:ea + nwin, ea:ea + nwin].permute(1, 2, 0).reshape(-1, 45)
    scale = zf.abs().max().item()
    assert (zp - inner).abs().max().item() <= 1e-12 * scale
    # oracle spot check
    rng = np.random.default_rng(5)
    rows, cols = rng.integers(0, H, 300), rng.integers(0, H, 300)
    ref = np.stack([zo.moments_frame_direct(frame, z.polynomials, rows=[r], cols=[c])[:, 0, 0]
                    for r, c in zip(rows, cols)])
    got = zf[:, torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)].T.cpu().numpy()
    rel_close(got, ref)
    # linearity of both kernels: Z(2 f + 3 g) = 2 Z(f) + 3 Z(g); float32 inputs chosen so the
    # combination is exact (small integers / 256)
    g1 = torch.randint(0, 64, (H, H), device=dev).float() / 256
    g2 = torch.randint(0, 64, (H, H), device=dev).float() / 256
    lhs = frame_moments_device(plan, 2 * g1 + 3 * g2)
    rhs = 2 * frame_moments_device(plan, g1) + 3 * frame_moments_device(plan, g2)
    assert (lhs - rhs).abs().max().item() <= 1e-12 * lhs.abs().max().item()
    del lhs, rhs, zf, inner
    p1 = torch.randint(0, 64, (100000, K, K), device=dev).float() / 256
    p2 = torch.randint(0, 64, (100000, K, K), device=dev).float() / 256
    lhs = patch_moments_device(plan, 2 * p1 + 3 * p2)
    rhs = 2 * patch_moments_device(plan, p1) + 3 * patch_moments_device(plan, p2)
    assert (lhs - rhs).abs().max().item() <= 1e-12 * lhs.abs().max().item()


def test_row_bands_reassemble_full_frame(native):
    """The multi-GPU row-band entry point: bands computed separately equal the full result."""
    torch = _torch()
    from mtflearn_amd.distributed import frame_moments_device, shard_bounds
    z = _zps(10, 32)
    plan = z._device_plan()
    dev = torch.device("cuda:0")
    img = torch.rand((203, 301), device=dev)
    full = frame_moments_device(plan, img)
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            start, count, padded = shard_bounds(203, rank, world)
            parts.append(frame_moments_device(plan, img, row0=start, n_rows=count))
        assert torch.equal(torch.cat(parts, dim=1), full)


def test_sharded_helpers_single_rank_group(native, zo):
    """The sharded drivers on the test-aid communicator (world-size-1 gloo group) on the GPU: same driver code
    as N ranks, checked against the oracle.  (The RCCL communicator: tests/test_gpu_runtime.py.)"""
    torch = _torch()
    import os
    import torch.distributed as dist
    from mtflearn_amd.distributed import TorchComm, sharded_patch_moments, sharded_frame_moments
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = TorchComm()
        z = _zps(8, 32)
        plan = z._device_plan()
        rng = np.random.default_rng(3)
        p = rng.random((333, 32, 32), dtype=np.float32)
        got = sharded_patch_moments(plan, comm, torch.from_numpy(p).cuda(), 333).cpu().numpy()
        rel_close(got, zo.moments_patches(p, z.polynomials))
        img = rng.random((45, 77), dtype=np.float32)
        gotf = sharded_frame_moments(plan, comm, torch.from_numpy(img).cuda()).cpu().numpy()
        rel_close(gotf, zo.moments_frame_direct(img, z.polynomials))
    finally:
        dist.destroy_process_group()


def test_config3_and_config5_sizes(native, zo):
    """Config 3 (64-px, n_max=12) and config 5 (n_max=10) at 4096 x 4096 would write 12.2 / 8.9 GB;
    here a 1024-row band of the 4096-wide frame exercises the same kernels and grid shapes, checked
    at random positions against the oracle."""
    torch = _torch()
    from mtflearn_amd.synthetic import honeycomb_frame
    from mtflearn_amd.distributed import frame_moments_device
    frame = honeycomb_frame(4096, seed=1)
    dev = torch.device("cuda:0")
    f = torch.from_numpy(frame).to(dev)
    rng = np.random.default_rng(9)
    for n_max, size in ((12, 64), (10, 32)):
        z = _zps(n_max, size)
        band = frame_moments_device(z._device_plan(), f, row0=1536, n_rows=1024)
        rows, cols = rng.integers(1536, 2560, 40), rng.integers(0, 4096, 40)
        ref = np.stack([zo.moments_frame_direct(frame, z.polynomials, rows=[r], cols=[c])[:, 0, 0]
                        for r, c in zip(rows, cols)])
        got = band[:, torch.from_numpy(rows - 1536).to(dev), torch.from_numpy(cols).to(dev)].T.cpu().numpy()
        rel_close(got, ref)
        del band
        if n_max == 10:
            # config 5 end to end: the fused maps of the same band against the reference's tail
            # (oracle rot_maps / to_complex / mirror_map) applied to the oracle moments of those pixels
            from mtflearn_amd.distributed import frame_maps_device
            theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
            n_c = sum(k // 2 + 1 for k in range(n_max + 1))
            rot, ab, mir = frame_maps_device(z._device_plan(), f, n_c, folds=(2, 3, 4, 6), theta=theta,
                                             row0=1536, n_rows=1024)
            ri, ci = torch.from_numpy(rows - 1536).to(dev), torch.from_numpy(cols).to(dev)
            rel_close(rot[:, ri, ci].T.cpu().numpy(), zo.rot_maps(ref, z.n, z.m, [2, 3, 4, 6]), rtol=1e-8,
                      atol_scale=1e-11)
            rel_close(ab[:, ri, ci].T.cpu().numpy(), np.abs(zo.to_complex(ref, z.n, z.m)[0]), rtol=1e-8,
                      atol_scale=1e-11)
            rel_close(mir[ri, ci].cpu().numpy(), zo.mirror_map(ref, z.n, z.m), rtol=1e-8, atol_scale=1e-11)
            del rot, ab, mir


@pytest.mark.parametrize("n_max,size,dtype", [(28, 56, np.float32), (25, 26, np.float64), (36, 72, np.float32)])
def test_large_sets_two_implementations_of_the_plain_sum(native, zo, n_max, size, dtype, monkeypatch):
    """n_max 25-40: the matrix-core kernels of zk_direct_patches.hip and the per-lane generic kernel (ZK_NO_DIRECT=1) are two
    independent implementations of the same unfolded sum over the caller's basis values; both must match the oracle, batch and
    dense, ragged counts and borders included."""
    rng = np.random.default_rng(n_max * 7 + size)
    z = _zps(n_max, size)
    plan = z._device_plan()
    assert not plan.has_path(0, native.ZK_F32, native.PATH_SEPARABLE)
    p = (rng.random((64 + 64 + 13, size, size)) - 0.4).astype(dtype)          # two whole waves and a ragged one
    img = (rng.random((size + 9, size + 70)) - 0.5).astype(dtype)
    ref_p = zo.moments_patches(p, z.polynomials)
    ref_f = zo.moments_frame_direct(img, _dense_basis(zo, z))
    got = {}
    for label, env in (("direct", None), ("per-lane", "1")):
        if env:
            monkeypatch.setenv("ZK_NO_DIRECT", env)
        else:
            monkeypatch.delenv("ZK_NO_DIRECT", raising=False)
        got[label] = (z.transform(p).data.copy(), z.transform(img).data.copy())
        rel_close(got[label][0], ref_p)
        rel_close(got[label][1], ref_f)
    monkeypatch.delenv("ZK_NO_DIRECT", raising=False)
    for a, b in zip(got["direct"], got["per-lane"]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-13 * np.abs(b).max())


@pytest.mark.parametrize("n_max,size,dtype,n_patches", [(20, 40, np.float32, 512 + 64 + 7), (19, 40, np.float64, 1100),
                                                        (13, 26, np.float32, 333), (24, 48, np.float32, 64)])
def test_direct_batch_kernel_ragged_batches_and_chunk_widths(native, zo, n_max, size, dtype, n_patches):
    """zk_patch_direct_kernel: four waves of 64 patches per workgroup walk the table together, so the last workgroup of a
    ragged batch holds whole waves past the end (they keep its barriers and store nothing) and a partly filled one; chunks of
    unequal width (231 functions = 5 + 5 + 5 blocks of 16, 210 = 5 + 5 + 4, 105 = 4 + 3, 325 = 6 + 5 + 5 + 5); NaNs outside
    the disk; rows behind the batch's results untouched."""
    import torch
    rng = np.random.default_rng(n_max + size + n_patches)
    z = _zps(n_max, size)
    plan = z._device_plan()
    plan.set_path(native.PATH_DIRECT)
    disk = np.any(z.polynomials != 0, axis=0)
    p = (rng.random((n_patches, size, size)) - 0.4).astype(dtype)
    dirty = p.copy()
    dirty[:, ~disk] = np.nan
    p_dev = torch.from_numpy(p).cuda()
    try:
        got = plan.transform_patches(p)
        np.testing.assert_array_equal(plan.transform_patches(dirty), got)
        out = torch.full((n_patches + 3, len(z.n)), -7.0, dtype=torch.float64, device="cuda")
        plan.transform_patches_dev(p_dev.data_ptr(), native.dtype_code(np.dtype(dtype)), n_patches, out.data_ptr())
        torch.cuda.synchronize()
        out = out.cpu().numpy()
        assert (out[n_patches:] == -7.0).all()
        np.testing.assert_array_equal(out[:n_patches], got)
    finally:
        plan.set_path(native.PATH_AUTO)
    rel_close(got, zo.moments_patches(p, z.polynomials))


def test_hbm_probe_reports_plausible_stream_rates(native):
    """zk_hbm_probe (bench.py roofline.this_box): read stream, copy and read stream with stores, on 1 GiB."""
    import torch
    src = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")                 # 1 GiB
    dst = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    n = src.numel() * 4
    ms_read = native.hbm_probe(0, src.data_ptr(), n)
    ms_copy = native.hbm_probe(0, src.data_ptr(), n, dst_ptr=dst.data_ptr())
    ms_mix = native.hbm_probe(0, src.data_ptr(), n, dst_ptr=dst.data_ptr(), store_per_group=23040)
    for ms in (ms_read, ms_copy, ms_mix):
        assert 0.05 < ms < 5.0                                                     # 0.2 .. 20 TB/s: a timing, not a constant
    assert ms_read <= ms_mix * 1.05
    with pytest.raises(RuntimeError, match="zk_hbm_probe"):
        native.hbm_probe(0, src.data_ptr(), 1000)


# ------------------------------------------------------------------ round 4: reference outputs on structured inputs
STRUCTURED_ORDERS = ((10, 32), (12, 64), (14, 32), (16, 32), (18, 40), (20, 40), (22, 48), (24, 48))


def _sample_index(n, step):
    return np.array(sorted(set(range(0, n, step)) | {n - 1}), dtype=np.int64)


@pytest.mark.parametrize("n_max,size", STRUCTURED_ORDERS)
def test_auto_path_against_reference_on_structured_inputs(native, golden, n_max, size):
    """The orders the reference's own estimator returns (12 .. size / 2, _estimate_n_max.py:95,123) on inputs where
    most moments are far below max|Z| -- the reference's test blobs and windows / a crop of a honeycomb lattice --
    against REFERENCE outputs (oracle/make_golden.py: st_*), by SURVEY 8c's criterion verbatim: elementwise rtol 1e-6
    with a floor of 1e-12 max|Z| up to n_max 12 and 1e-11 above.  ZK_PATH_AUTO, i.e. what ZPs.transform runs: the
    polynomial kernels up to n_max 16, the plain sum on the matrix cores above."""
    tag = f"{n_max}_{size}"
    z = _zps(n_max, size)
    plan = z._device_plan()
    floor = 1e-12 if n_max <= 12 else 1e-11
    batch, ref = np.ascontiguousarray(golden[f"st_batch_{tag}"]), golden[f"st_Z_{tag}"]
    assert np.median(np.abs(ref)) < 0.05 * np.abs(ref).max()
    rel_close(z.transform(batch).data, ref, atol_scale=floor)                      # 24-28 patches: less than a wave
    big = np.ascontiguousarray(np.concatenate([batch] * 5))                        # whole waves + a ragged one
    got = z.transform(big).data
    for k in range(5):
        rel_close(got[k * len(batch):(k + 1) * len(batch)], ref, atol_scale=floor)
    f32, f64 = native.ZK_F32, native.ZK_F64
    if n_max >= 17:
        assert plan.best_path(0, f32, len(big)) == native.PATH_DIRECT and plan.best_path(1, f64) == native.PATH_DIRECT
        assert plan.best_path(0, f32, len(batch)) == native.PATH_GENERIC
    else:
        assert plan.best_path(0, f32, len(big)) in (native.PATH_SEPARABLE, native.PATH_STREAM)
        assert plan.best_path(1, f64) == native.PATH_SEPARABLE
    crop = golden[f"st_frame_{tag}"].astype(np.float64)                            # float64 cast: the FFT oracle is exact
    H, W = crop.shape
    dense = z.transform(crop).data
    rel_close(dense[:, _sample_index(H, 4)][:, :, _sample_index(W, 5)], golden[f"st_Zf_{tag}"], atol_scale=floor)
    mx = float(golden[f"st_Zf_max_{tag}"])
    np.testing.assert_allclose(dense.sum(axis=(1, 2)), golden[f"st_Zf_sum_{tag}"], rtol=1e-9, atol=floor * H * W * mx)
    # the float32 crop itself: the same numbers (the cast is exact)
    np.testing.assert_array_equal(z.transform(golden[f"st_frame_{tag}"]).data, dense)
    # every family that sums the caller's own numbers meets the criterion at every order; the polynomial kernels,
    # forced, keep their documented distance
    for mode, arr, want in ((0, big, np.concatenate([ref] * 5)), (1, crop, None)):
        for name, res in _both_paths(native, z, arr, mode).items():
            if mode == 1:
                res, want = res[:, _sample_index(H, 4)][:, :, _sample_index(W, 5)], golden[f"st_Zf_{tag}"]
            rel_close(res, want, atol_scale=_floor(name, n_max))


@pytest.mark.parametrize("n_max,size", [(28, 56), (32, 64), (36, 72)])
def test_high_orders_against_reference(native, golden_high, n_max, size):
    """n_max 28 / 32 / 36 on 56 / 64 / 72-px windows (what the reference's estimator returns there) against REFERENCE outputs
    on structured inputs (oracle/make_golden_high_orders.py: hi_*), SURVEY 8c's criterion verbatim with the floor of 1e-11
    max|Z|: ZK_PATH_AUTO = the plain sum on the matrix cores, batch (whole waves and a ragged one) and dense (float64-cast
    crop, strided positions incl. the zero-padded borders, per-plane sums); fewer than 64 patches = the per-lane kernel."""
    g, tag = golden_high, f"{n_max}_{size}"
    z = _zps(n_max, size)
    plan = z._device_plan()
    batch, ref = np.ascontiguousarray(g[f"hi_batch_{tag}"]), g[f"hi_Z_{tag}"]
    rel_close(z.transform(batch).data, ref, atol_scale=1e-11)                       # 14 patches: the per-lane kernel
    big = np.ascontiguousarray(np.concatenate([batch] * 10))                        # 140 patches: two waves and a ragged one
    assert plan.best_path(0, native.ZK_F32, len(big)) == native.PATH_DIRECT and plan.best_path(1, native.ZK_F64) == native.PATH_DIRECT
    got = z.transform(big).data
    for k in range(10):
        rel_close(got[k * len(batch):(k + 1) * len(batch)], ref, atol_scale=1e-11)
    crop = g[f"hi_frame_{tag}"].astype(np.float64)
    H, W = crop.shape
    dense = z.transform(crop).data
    rel_close(dense[:, _sample_index(H, 8)][:, :, _sample_index(W, 9)], g[f"hi_Zf_{tag}"], atol_scale=1e-11)
    mx = float(g[f"hi_Zf_max_{tag}"])
    np.testing.assert_allclose(dense.sum(axis=(1, 2)), g[f"hi_Zf_sum_{tag}"], rtol=1e-9, atol=1e-11 * H * W * mx)
    np.testing.assert_array_equal(z.transform(g[f"hi_frame_{tag}"]).data, dense)   # the float32 crop: the cast is exact


def test_config0_frame_against_reference(native, golden):
    """configs[0]: the reference's own 512 x 512 test image (datasets/_zps_test_data.py:62-65, seed 0), 32-px patches,
    n_max 8 -- the batch path on a strided grid of windows and the dense path on the whole frame against reference outputs."""
    z = _zps(8, 32)
    frame = golden["c0_frame_512"]
    grid = golden["c0_grid"]
    win = np.ascontiguousarray(np.array([frame[r:r + 32, c:c + 32] for r in grid for c in grid]))
    zb = z.transform(win).data
    rel_close(zb, golden["c0_Z_grid_8_32"])
    np.testing.assert_allclose(zb, golden["c0_Z_grid_8_32"], rtol=1e-6, atol=0)      # no floor at all
    got_grid = z.transform_grid(frame, 16).data                                       # the reference's strided extractor
    rel_close(got_grid, golden["c0_Z_grid_8_32"])
    ri = golden["c0_sample_index"]
    mx = float(golden["c0_Zf_f64cast_max"])
    dense = z.transform(frame.astype(np.float64)).data
    assert dense.shape == (45, 512, 512)
    rel_close(dense[:, ri][:, :, ri], golden["c0_Zf_f64cast_sample"])
    np.testing.assert_allclose(dense.sum(axis=(1, 2)), golden["c0_Zf_f64cast_sum"], rtol=1e-9, atol=1e-12 * 512 * 512 * mx)
    np.testing.assert_allclose(np.abs(dense).sum(axis=(1, 2)), golden["c0_Zf_f64cast_abs_sum"], rtol=1e-9)
    d32 = z.transform(frame).data                                                     # float32 in: exact cast -> same numbers
    np.testing.assert_array_equal(d32, dense)
    # the reference on the float32 image runs its FFT in single precision: norm-wise 1e-6
    assert np.abs(d32[:, ri][:, :, ri] - golden["c0_Zf_f32_sample"]).max() <= 1e-6 * mx
    np.testing.assert_array_equal(z.transform(frame).valid_mask[ri][:, ri],
                                  np.pad(np.ones((512 - 31, 512 - 31), bool), ((15, 16), (15, 16)))[ri][:, ri])


def test_fused_symmetry_maps_against_reference_on_a_structured_crop(native, golden):
    """configs[4]'s parameters (n_max 10, 32-px windows): the fused frame -> maps kernel against the REFERENCE's own tail
    (zmoments.rot_maps / to_complex / mirror_map of its dense moments, _zmoments.py:300-316, 420-493) on a honeycomb crop --
    float64-cast in the reference, float32 and float64 in, here; strided positions incl. the zero-padded borders."""
    z = _zps(10, 32)
    crop = golden["st_maps_frame_10_32"]
    ri, ci = _sample_index(46, 3), _sample_index(58, 4)
    pick = lambda a: a[..., ri, :][..., ci]
    for img in (crop, crop.astype(np.float64)):
        maps = z.symmetry_maps(img)
        np.testing.assert_allclose(pick(maps["rot_maps"]), golden["st_maps_rot_10_32"], rtol=1e-8, atol=1e-11)
        rel_close(pick(maps["abs"]), golden["st_maps_abs_10_32"], rtol=1e-8)
        np.testing.assert_allclose(pick(maps["mirror_map"]), golden["st_maps_mirror_10_32"], rtol=1e-8, atol=1e-11)
        got = z.symmetry_maps(img, n_folds=[3, 5], p=None, m_unselect=(0, 1, 2), mirror=False, abs_moments=False)
        np.testing.assert_allclose(pick(got["rot_maps"]), golden["st_maps_rot_pnone_unsel012_10_32"], rtol=1e-8,
                                   atol=1e-12 * np.abs(golden["st_maps_rot_pnone_unsel012_10_32"]).max())
        # the host container on the device moments: the same reference numbers
        zm = z.transform(img)
        np.testing.assert_allclose(pick(zm.rot_maps([2, 3, 4, 6])), golden["st_maps_rot_10_32"], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(pick(zm.mirror_map()), golden["st_maps_mirror_10_32"], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("n_max,size", [(20, 40), (28, 56)])
def test_symmetry_maps_at_high_orders_against_reference(native, golden_high, n_max, size):
    """ZPs.symmetry_maps where the moments come from the matrix-core kernel (n_max 20: a band of them feeds the class-pass tail;
    28: the planes tail) against the REFERENCE's own rot_maps / |to_complex| / mirror_map of its dense moments of a honeycomb crop
    (oracle/make_golden_high_orders.py: hi_maps_*), float32 and float64 in."""
    g, tag = golden_high, f"{n_max}_{size}"
    z = _zps(n_max, size)
    crop = g[f"hi_maps_frame_{tag}"]
    ri, ci = _sample_index(crop.shape[0], 6), _sample_index(crop.shape[1], 7)
    pick = lambda a: a[..., ri, :][..., ci]
    for img in (crop, crop.astype(np.float64)):
        maps = z.symmetry_maps(img)
        np.testing.assert_allclose(pick(maps["rot_maps"]), g[f"hi_maps_rot_{tag}"], rtol=1e-8, atol=1e-11)
        rel_close(pick(maps["abs"]), g[f"hi_maps_abs_{tag}"], rtol=1e-8, atol_scale=1e-11)
        np.testing.assert_allclose(pick(maps["mirror_map"]), g[f"hi_maps_mirror_{tag}"], rtol=1e-8, atol=1e-11)


def test_keypoints_moments_against_the_reference(native):
    """mtflearn_amd.features.KeyPoints (the reference's class, features/_keypoint.py:53-92): border clearing + windows as the
    reference cuts them, and `moments(zps)` -- the windows read from the resident frame on the GPU -- against the REFERENCE's
    ZPs.transform of its own extracted patches (oracle/make_golden_keypoints.py), even and odd window sizes."""
    import os
    from conftest import ROOT
    from mtflearn_amd.features import KeyPoints
    with np.load(os.path.join(ROOT, "tests", "golden", "keypoints_golden.npz")) as f:
        kg = {k: f[k] for k in f.files}
    for size in (32, 33):
        kp = KeyPoints(kg["pts"], kg["frame"], size)
        z = _zps(8, size)
        zm = kp.moments(z)
        assert zm.data.shape == kg[f"Z_{size}"].shape and zm.patch_size == size
        rel_close(zm.data, kg[f"Z_{size}"])
        rel_close(z.transform(kp.extract_patches()).data, kg[f"Z_{size}"])          # the batch route: same numbers


def test_keypoints_moments_at_a_high_order_against_the_reference(native, golden_high):
    """The same at n_max 20 on 40-px windows (123 points kept of 260): `KeyPoints.moments` gathers the windows on the device and runs
    the matrix-core plain sum; against the REFERENCE's ZPs(20, 40).transform of the windows its own KeyPoints cut."""
    from mtflearn_amd.features import KeyPoints
    g = golden_high
    kp = KeyPoints(g["hi_kp_pts"], g["hi_kp_frame"], 40)
    np.testing.assert_array_equal(kp.pts, g["hi_kp_kept_40"])
    z = _zps(20, 40)
    zm = kp.moments(z)
    assert zm.data.shape == g["hi_kp_Z_20_40"].shape
    rel_close(zm.data, g["hi_kp_Z_20_40"], atol_scale=1e-11)
    rel_close(z.transform(kp.extract_patches()).data, g["hi_kp_Z_20_40"], atol_scale=1e-11)
    assert z._device_plan().best_path(0, native.ZK_F32, len(kp.pts)) == native.PATH_DIRECT


def test_auto_never_takes_the_polynomial_kernels_above_n_max_16(native, zo):
    """A window too large for the matrix-core dense kernel's LDS tile (float64, 106 px: 155 KB) but not for the separable kernel's
    (147 KB), at an order AUTO serves with the plain sum: the per-lane plain sum runs instead of the separable family -- slow, and
    exact (the contract of INTEGRATION section 4)."""
    z = _zps(18, 106)
    plan = z._device_plan()
    assert not plan.has_path(1, native.ZK_F64, native.PATH_DIRECT) and plan.has_path(1, native.ZK_F64, native.PATH_SEPARABLE)
    assert plan.best_path(1, native.ZK_F64) == native.PATH_GENERIC
    assert plan.best_path(0, native.ZK_F64, 1000) == native.PATH_DIRECT and plan.best_path(0, native.ZK_F64, 10) == native.PATH_GENERIC
    rng = np.random.default_rng(3)
    img = rng.random((110, 120))
    ref = zo.moments_frame_direct(img, _dense_basis(zo, z), rows=[0, 53, 109], cols=[0, 64, 119])
    got = z.transform(img).data[:, [0, 53, 109]][:, :, [0, 64, 119]]
    rel_close(got, ref, atol_scale=1e-11)
