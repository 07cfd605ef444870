"""The key-point caller (mtflearn_amd.features.KeyPoints) against outputs of the reference's own
features/_keypoint.py (oracle/make_golden_keypoints.py -> tests/golden/keypoints_golden.npz).  No GPU."""
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def kg():
    with np.load(os.path.join(ROOT, "tests", "golden", "keypoints_golden.npz")) as f:
        return {k: f[k] for k in f.files}


def test_keypoints_match_the_reference(kg):
    from mtflearn_amd.features import KeyPoints
    from mtflearn_amd.features.keypoints import clear_border, center_of_mass_refine, disk_patch
    frame, pts = kg["frame"], kg["pts"]
    for size in (32, 33):
        k = KeyPoints(pts, frame, size)
        np.testing.assert_array_equal(k.pts, kg[f"kept_{size}"])
        np.testing.assert_array_equal(clear_border(pts, frame.shape, size), kg[f"clear_border_{size}"])
        assert k.shape == frame.shape and k.size == size and k.patches is None
        patches = k.extract_patches()
        assert patches is k.patches and patches.dtype == frame.dtype
        np.testing.assert_array_equal(patches, kg[f"patches_{size}"])
        np.testing.assert_array_equal(k.extract_patches(flat=True)[:3], kg[f"patches_flat_{size}_head"])
        k.clear_border(48)                                             # y bounded by the WIDTH, as the reference does it
        np.testing.assert_array_equal(k.pts, kg[f"kept_after_48_{size}"])
    k = KeyPoints(pts, frame, 24)
    np.testing.assert_array_equal(k.extract_patches(16), kg["patches_16_of_24"])
    k.refine(r=3)
    np.testing.assert_allclose(k.pts, kg["refined_r3"], rtol=1e-12)
    k2 = KeyPoints(pts, frame, 24)
    k2.refine(r=4, mode='disk')
    np.testing.assert_allclose(k2.pts, kg["refined_r4_disk"], rtol=1e-12)
    np.testing.assert_array_equal(disk_patch(5), kg["disk_5"])
    ipts = np.rint(clear_border(pts, frame.shape, 24)).astype(int)[:40]
    np.testing.assert_allclose(center_of_mass_refine(frame, ipts, size=2), kg["com_refine_box"], rtol=1e-12)


def test_windows_outside_the_frame_fall_back_to_numpy_slicing():
    """Without border clearing (points set by hand) the batch is whatever NumPy's slices give, as in the reference: a window
    that runs off the frame comes back short, and NumPy refuses the ragged batch."""
    from mtflearn_amd.features import KeyPoints
    img = np.arange(100.0).reshape(10, 10)
    k = KeyPoints(np.array([[5.0, 5.0]]), img, 2)
    assert k.pts.shape == (1, 2)
    k.pts = np.array([[5.0, 5.0], [4.0, 6.0]])
    np.testing.assert_array_equal(k.extract_patches(4), np.array([img[3:7, 3:7], img[4:8, 2:6]]))
    k.pts = np.array([[5.0, 5.0], [10.0, 10.0]])                       # the second window is img[8:12, 8:12] -> 2 x 2
    with pytest.raises(ValueError):
        k.extract_patches(4)
