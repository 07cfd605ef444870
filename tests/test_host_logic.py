"""Host side of the drop-in boundary (no GPU): ZPs construction and validation, the zmoments
container and index algebra, and the C-ABI library's exported surface.

The nm2j / select cases restate the reference's own tests (tests/features/test_zmoments.py:5-88)
against the product names, so they read like the reference's suite."""
import ctypes
import os
import pickle
import re
import warnings

import numpy as np
import pytest
from sklearn.base import clone

from conftest import ROOT
import mtflearn_amd
from mtflearn_amd import ZPs, zmoments, _native
from mtflearn_amd.features import (nm2j, nm2j_complex, construct_complex_matrix, construct_real_matrix,
                                   construct_rot_maps_matrix, check_array1d)


# ---------------------------------------------------------------- reference test_zmoments.py
def test_scalar_inputs():
    assert nm2j(0, 0) == 0
    assert nm2j(1, -1) == 1
    assert nm2j(2, 0) == 4
    assert nm2j(3, 1) == 8
    assert nm2j(4, -4) == 10
    assert nm2j(5, 3) == 19


def test_array_inputs():
    np.testing.assert_array_equal(nm2j([0, 1, 2, 2, 3], [0, -1, 0, 2, 3]), [0, 1, 4, 5, 9])


def test_edge_cases():
    assert nm2j(0, 0) == 0
    big = 1000
    assert nm2j(big, big) == ((big + 2) * big + big) // 2


def test_invalid_inputs():
    with pytest.raises(ValueError, match="Radial order `n` must be non-negative."):
        nm2j(-1, 0)
    with pytest.raises(ValueError, match="Azimuthal frequency `m` must satisfy \\|m\\| ≤ n."):
        nm2j(2, 3)
    with pytest.raises(ValueError):
        nm2j(0, 1)
    with pytest.raises(ValueError, match="`n - \\|m\\|` must be even."):
        nm2j(1, 0)
    with pytest.raises(ValueError):
        nm2j(3, 2)
    with pytest.raises(ValueError, match="`n` and `m` must have the same shape."):
        nm2j([1, 2], [0])
    assert nm2j(2.0, 0.0) == 4
    with pytest.raises(ValueError):
        nm2j(2.5, 0)
    with pytest.raises(ValueError):
        nm2j(2, 0.5)


def test_large_array_and_output_type():
    n = np.arange(0, 100) * 2
    m = np.where(np.arange(100) % 2 == 0, 0, 1) * 2
    j = nm2j(n, m)
    assert len(j) == 100 and j[0] == 0 and j[1] == ((n[1] + 2) * n[1] + m[1]) // 2
    assert isinstance(nm2j(2, 0), int)
    assert isinstance(nm2j([2], [0]), np.ndarray)


def test_zmoments_select_filters_by_absolute_m_values():
    data = np.arange(12, dtype=float).reshape(2, 6)
    n = np.array([0, 1, 1, 2, 2, 3])
    m = np.array([0, -1, 1, -2, 2, 3])
    sel = zmoments(data=data, n=n, m=m).select([1, -2])
    np.testing.assert_array_equal(sel.m, np.array([-1, 1, -2, 2]))
    np.testing.assert_array_equal(sel.n, np.array([1, 1, 2, 2]))
    np.testing.assert_array_equal(sel.data, data[:, [1, 2, 3, 4]])


# ---------------------------------------------------------------- ZPs host behaviour
def test_zps_constructor_contract(golden):
    z = ZPs(8, 32)
    assert repr(z) == "ZPs(n_max=8, size=32)"
    assert z.get_params() == {"n_max": 8, "size": 32}
    assert clone(z).get_params() == z.get_params()
    assert z.polynomials.dtype == np.float64 and z.polynomials.shape == (45, 32, 32)
    assert np.array_equal(z.polynomials, golden["basis_8_32"])       # bit-identical to the reference
    assert np.array_equal(z.n, golden["n_8"]) and np.array_equal(z.m, golden["m_8"])
    assert z.get_polynomials() is z.polynomials
    assert z.fit(None) is z
    z2 = pickle.loads(pickle.dumps(z))
    assert np.array_equal(z2.polynomials, z.polynomials)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for key, (n_max, size) in {"basis_5_9": (5, 9), "basis_10_11": (10, 11), "basis_4_8": (4, 8)}.items():
            assert np.array_equal(ZPs(n_max, size).polynomials, golden[key]), key
    assert np.array_equal(ZPs(12, 64).polynomials[:, ::7, ::5], golden["basis_12_64_sample"])


def test_zps_validation_messages():
    with pytest.raises(ValueError, match="n_max must be non-negative."):
        ZPs(-1, 8)
    with pytest.raises(ValueError, match="size must be positive."):
        ZPs(2, 0)
    with pytest.raises(ValueError, match=r"n_max=9 exceeds size=8\. This will produce meaningless results\. "
                                         r"Use n_max <= 4 for accurate moments\."):
        ZPs(9, 8)
    with pytest.warns(UserWarning, match=r"n_max=5 exceeds recommended limit of size/2≈4\."):
        ZPs(5, 8)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ZPs(4, 8)                                                      # exactly size/2: no warning
    z = ZPs(4, 8)
    with pytest.raises(ValueError, match="Images must be 2D or 3D array."):
        z.transform(np.zeros(8))
    with pytest.raises(ValueError, match=r"For batch processing, image size \(7x8\) must match polynomial size \(8x8\)"):
        z.transform(np.zeros((2, 7, 8)))
    with pytest.raises(ValueError, match=r"For FFT convolution, image size \(7x20\) must be at least as large as "
                                         r"polynomial size \(8x8\)"):
        z.transform(np.zeros((7, 20)))
    empty = z.transform(np.zeros((0, 8, 8), dtype=np.float32))       # no device needed for N == 0
    assert empty.data.shape == (0, 15) and empty.data.dtype == np.float64


def test_extension_methods_validate_before_touching_the_device():
    z = ZPs(4, 8)
    with pytest.raises(ValueError, match="symmetry_maps needs a 2D image."):
        z.symmetry_maps(np.zeros((2, 8, 8)))
    with pytest.raises(ValueError, match=r"image size \(7x20\) must be at least"):
        z.symmetry_maps(np.zeros((7, 20)))
    with pytest.raises(ValueError, match="m=0 must be included in m_unselect."):
        z.symmetry_maps(np.zeros((16, 16)), m_unselect=(1, 2))
    with pytest.raises(ValueError, match="transform_at needs a 2D image."):
        z.transform_at(np.zeros(8), [[1, 1]])
    empty = z.transform_at(np.zeros((16, 16)), np.empty((0, 2)))
    assert empty.data.shape == (0, 15)
    np.testing.assert_array_equal(z._valid_mask(20, 23), zmoments(np.zeros((15, 20, 23)), z.n, z.m, 8).valid_mask)


# ---------------------------------------------------------------- container vs reference vectors
def test_container_matches_reference(golden):
    z8 = ZPs(8, 32)
    zb = zmoments(golden["Z_blobs_8_32"], z8.n, z8.m, patch_size=32)
    assert zb.data is not None and zb.valid_mask is None and not zb.is_complex
    zc = zb.to_complex()
    np.testing.assert_array_equal(zc.data, golden["pp2_complex"])
    np.testing.assert_array_equal(zc.n, golden["pp2_complex_n"])
    np.testing.assert_array_equal(zc.m, golden["pp2_complex_m"])
    assert zc.is_complex and zc.to_complex() is zc and zb.to_real() is zb
    np.testing.assert_allclose(zb.rot_maps([2, 3, 4, 6]), golden["pp2_rot_maps"], rtol=1e-13)
    np.testing.assert_allclose(zb.rot_maps([3, 6], p=None), golden["pp2_rot_maps_pnone"], rtol=1e-13)
    np.testing.assert_allclose(zb.rot_maps([3], m_unselect=(0, 1, 2)), golden["pp2_rot_maps_unsel012"], rtol=1e-13)
    with pytest.raises(ValueError, match="m=0 must be included in m_unselect."):
        zb.rot_maps([3], m_unselect=(1, 2))
    np.testing.assert_allclose(zb.mirror_map(), golden["pp2_mirror"], rtol=1e-13)
    np.testing.assert_allclose(zb.normalize(order=2).data, golden["pp2_norm2"], rtol=1e-15)
    np.testing.assert_allclose(zb.normalize().data, golden["pp2_norm_none"], rtol=1e-15)
    np.testing.assert_allclose(zb.rotate(30.0).data, golden["pp2_rotate30"], rtol=1e-14)
    back = zc.to_real()
    np.testing.assert_array_equal(back.data, golden["pp2_toreal"])
    np.testing.assert_array_equal(back.n, golden["pp2_toreal_n"])
    np.testing.assert_array_equal(back.m, golden["pp2_toreal_m"])
    sel = zb.select([1, -2])
    np.testing.assert_array_equal(sel.data, golden["pp2_select"])
    np.testing.assert_array_equal(sel.n, golden["pp2_select_n"])
    np.testing.assert_array_equal(sel.m, golden["pp2_select_m"])
    np.testing.assert_array_equal(zb.unselect([0, 1]).m, golden["pp2_unselect_m"])
    # rank 3
    z6 = ZPs(6, 12)
    zf = zmoments(golden["pp3_moments"], z6.n, z6.m, patch_size=12)
    np.testing.assert_array_equal(zf.valid_mask, golden["pp3_valid_mask"])
    np.testing.assert_array_equal(zf.to_complex().data, golden["pp3_complex"])
    np.testing.assert_allclose(zf.rot_maps([2, 3, 4, 6]), golden["pp3_rot_maps"], rtol=1e-12)
    np.testing.assert_allclose(zf.mirror_map(), golden["pp3_mirror"], rtol=1e-12)
    np.testing.assert_allclose(zf.rotate(45.0).data, golden["pp3_rotate45"], rtol=1e-14)
    np.testing.assert_array_equal(zf.to_complex().to_real().data, golden["pp3_toreal"])
    np.testing.assert_allclose(zf.normalize(order=2).data, golden["pp3_norm2"], rtol=1e-15)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z9 = ZPs(5, 9)
    z59 = zmoments(np.zeros((21, 20, 23)), z9.n, z9.m, patch_size=9)
    np.testing.assert_array_equal(z59.valid_mask, golden["valid_mask_9_20_23"])


def test_container_validation_and_sorting():
    with pytest.raises(ValueError, match="`n` and `m` must have the same shape."):
        zmoments(np.zeros((2, 3)), [0, 1, 1], [0, -1])
    with pytest.raises(ValueError, match="Data shape mismatch: expected 3 moments but got 4"):
        zmoments(np.zeros((2, 4)), [0, 1, 1], [0, -1, 1])
    with pytest.raises(ValueError, match="Data shape mismatch: expected 3 moments but got 2"):
        zmoments(np.zeros((2, 4, 4)), [0, 1, 1], [0, -1, 1])
    with pytest.raises(ValueError, match="Data must be 2D or 3D array."):
        zmoments(np.zeros(3), [0, 1, 1], [0, -1, 1])
    data = np.arange(6.0).reshape(2, 3)
    z = zmoments(data, n=[1, 0, 1], m=[1, 0, -1])                      # unsorted labels are canonicalised
    np.testing.assert_array_equal(z.n, [0, 1, 1])
    np.testing.assert_array_equal(z.m, [0, -1, 1])
    np.testing.assert_array_equal(z.data, data[:, [1, 2, 0]])
    assert zmoments(np.zeros((3, 4, 5)), [0, 1, 1], [0, -1, 1]).valid_mask is None    # no patch_size


def test_index_algebra_matches_reference(golden):
    n, m = golden["n_8"], golden["m_8"]
    np.testing.assert_array_equal(construct_complex_matrix(n, m), golden["cmat_8"])
    inv, nr, mr = construct_real_matrix(golden["pp2_complex_n"], golden["pp2_complex_m"])
    np.testing.assert_array_equal(inv, golden["rmat_8"])
    np.testing.assert_array_equal(nr, golden["rmat_8_n"])
    np.testing.assert_array_equal(mr, golden["rmat_8_m"])
    np.testing.assert_array_equal(construct_rot_maps_matrix([1, 2, 3, 4, 6], m), golden["rotmat_8"])
    np.testing.assert_array_equal(nm2j(n, m), golden["nm2j_8"])
    np.testing.assert_array_equal(nm2j_complex(golden["pp2_complex_n"], golden["pp2_complex_m"]),
                                  golden["nm2j_complex_8"])
    assert nm2j_complex(4, 2) == 7 and isinstance(nm2j_complex(4, 2), int)
    with pytest.raises(ValueError, match="Azimuthal frequency m must be non-negative."):
        nm2j_complex(2, -2)
    np.testing.assert_array_equal(check_array1d(3), [3])
    np.testing.assert_array_equal(check_array1d([[1, 2], [3, 4]]), [1, 2, 3, 4])


# ---------------------------------------------------------------- C ABI surface (no compute)
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "zernike_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z_0-9]+)\s*\(", text)))


def test_device_operand_dtypes():
    """Detector formats (uint8 / uint16 / int16, bool and int8 re-labelled) travel as they are and are widened on the
    device (exact in float32); float16 goes to float32, wide integers to float64, as NumPy's promotion would give."""
    from mtflearn_amd import ZPs, _native
    for dt, want in [(np.uint8, np.uint8), (np.int16, np.int16), (np.uint16, np.uint16), (np.bool_, np.uint8),
                     (np.int8, np.int16), (np.float16, np.float32), (np.int32, np.float64), (np.uint32, np.float64),
                     (np.int64, np.float64), (np.float32, np.float32), (np.float64, np.float64)]:
        a = (np.arange(24).reshape(4, 6) % 2 if dt == np.bool_ else np.arange(24).reshape(4, 6) * 997 % 251 - (100 if dt == np.int8 else 0)).astype(dt)
        for view in (a, a[:, ::2]):
            op = ZPs._device_operand(view)
            assert op.dtype == want and op.flags.c_contiguous and _native.dtype_code(op.dtype) is not None
            np.testing.assert_array_equal(op.astype(np.float64), view.astype(np.float64))
    with pytest.raises(TypeError):
        ZPs._device_operand(np.zeros((2, 2), np.complex64))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/zernike_hip.h but not exported"
    assert sorted(_native.SYMBOLS) == declared                          # the ctypes table binds all of them
    assert _native.load().zk_abi_version() == 2


def test_product_fails_loudly_without_device():
    if _native.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        ZPs(4, 8).transform(np.zeros((2, 8, 8), dtype=np.float32))
    with pytest.raises(RuntimeError, match="no HIP device"):
        ZPs(4, 8).transform(np.zeros((16, 16), dtype=np.float32))
    # the consumers too: nothing computes on the CPU (the deviation from SURVEY 8b)
    from mtflearn_amd.features import pca
    from mtflearn_amd.clustering import kmeans_lbs
    X = np.random.default_rng(0).standard_normal((50, 5))
    with pytest.raises(RuntimeError):
        pca(X, 2)
    with pytest.raises(RuntimeError):
        kmeans_lbs(X, 3)
    # what does work without a device: the basis, validation, the container's host methods
    z = ZPs(4, 8)
    assert z.polynomials.shape == (15, 8, 8)
    zm = zmoments(np.ones((3, 15)), z.n, z.m)
    assert zm.to_complex().data.shape == (3, 9) and zm.rot_maps([2, 3]).shape == (3, 2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "motif-learn_amd", "mtflearn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "zernike_oracle" not in src, f


# ---------------------------------------------------------------- randomized cross-check with the oracle
@pytest.mark.parametrize("seed", range(6))
def test_container_matches_oracle_on_random_moments(seed):
    """The product container (features/moments.py) and the oracle (oracle/zernike_oracle.py) are two
    independent restatements of reference _zmoments.py; on random moments, random label order and random
    options they must agree to rounding."""
    from oracle import zernike_oracle as zo
    rng = np.random.default_rng(seed)
    n_max = int(rng.integers(2, 9))
    n = np.array([a for a in range(n_max + 1) for _ in range(-a, a + 1, 2)])
    m = np.array([b for a in range(n_max + 1) for b in range(-a, a + 1, 2)])
    perm = rng.permutation(len(n))                       # labels arrive unsorted
    shape = (7, len(n)) if seed % 2 == 0 else (len(n), 5, 6)
    data = rng.standard_normal(shape)
    axis = 1 if data.ndim == 2 else 0
    zm = zmoments(np.take(data, perm, axis=axis), n[perm], m[perm], patch_size=9)
    sd, sn, sm = zo.sort_by_nm(np.take(data, perm, axis=axis), n[perm], m[perm])
    np.testing.assert_array_equal(zm.data, sd)
    np.testing.assert_array_equal(zm.n, sn)
    np.testing.assert_array_equal(zm.m, sm)
    np.testing.assert_array_equal(zm.data, data)         # canonical order restored

    zc, nc, mc = zo.to_complex(data, n, m)
    got = zm.to_complex()
    np.testing.assert_allclose(got.data, zc, rtol=1e-14)
    np.testing.assert_array_equal(got.n, nc)
    np.testing.assert_array_equal(got.m, mc)
    np.testing.assert_allclose(got.to_real().data, data, rtol=1e-14, atol=1e-15)

    folds = sorted(set(int(v) for v in rng.integers(1, 8, size=3)))
    unsel = (0, 1) if seed % 3 else (0, 1, 2)
    p = [2, None, 1][seed % 3]
    np.testing.assert_allclose(zm.rot_maps(folds, p=p, m_unselect=unsel),
                               zo.rot_maps(data, n, m, folds, p=p, m_unselect=unsel), rtol=1e-12, atol=1e-14)
    theta = np.linspace(0, 2 * np.pi, 24, endpoint=False) if seed % 2 else None
    np.testing.assert_allclose(zm.mirror_map(theta=theta, p=p, m_unselect=unsel),
                               zo.mirror_map(data, n, m, theta=theta, p=p, m_unselect=unsel), rtol=1e-12, atol=1e-14)
    ang = float(rng.uniform(-180, 180))
    np.testing.assert_allclose(zm.rotate(ang).data, zo.rotate(data, n, m, ang)[0], rtol=1e-13, atol=1e-15)
    pick = [int(v) for v in rng.integers(-n_max, n_max + 1, size=2)]
    sel = zm.select(pick)
    d2, n2, m2 = zo.select(data, n, m, pick)
    np.testing.assert_array_equal(sel.data, d2)
    np.testing.assert_array_equal(sel.m, m2)
    np.testing.assert_allclose(zm.normalize(order=p).data, zo.normalize(data, order=p), rtol=1e-14)
    if data.ndim == 3:
        np.testing.assert_array_equal(zm.valid_mask, zo.valid_mask(data.shape[1:], 9))


def test_pca_host_arithmetic_matches_sklearn():
    """The host half of the device pca (covariance from the Gram matrix, eigh, ordering, sign convention) against
    scikit-learn's PCA -- the reference's pca() is PCA(n).fit_transform(X) (features/_dimension_reduction.py:3-6).
    The two device passes (zk_gram, zk_project) are stood in for by NumPy here; tests/test_gpu_consumers.py runs them."""
    from sklearn.decomposition import PCA
    from mtflearn_amd.features.consumers import _covariance_eigh
    rng = np.random.default_rng(4)
    X = rng.standard_normal((3000, 45)) * rng.random(45) * 2 + rng.random(45)
    aug = np.hstack([X, np.ones((X.shape[0], 1))])
    mean, vt, var = _covariance_eigh(aug.T @ aug, X.shape[0])
    model = PCA(n_components=6).fit(X)
    np.testing.assert_allclose(mean, model.mean_, rtol=1e-12)
    np.testing.assert_allclose(vt[:6], model.components_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(var[:6], model.explained_variance_, rtol=1e-9)
    np.testing.assert_allclose((X - mean) @ vt[:6].T, model.transform(X), rtol=0, atol=1e-9)


def test_pinned_pool_disabled_hands_out_plain_arrays(monkeypatch):
    monkeypatch.setenv("MTFLEARN_AMD_PINNED_MB", "0")
    pool = _native.PinnedPool()
    a = pool.empty((1024, 1024))
    assert a.flags.owndata and a.shape == (1024, 1024) and a.dtype == np.float64
    small = _native.PinnedPool().empty((4, 4))            # below the size where a pinned block pays off
    assert small.flags.owndata
