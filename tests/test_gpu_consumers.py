"""pca on the device against scikit-learn's PCA, which is what the reference's pca() calls."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pca_matches_sklearn_on_a_moment_matrix():
    from sklearn.decomposition import PCA
    from mtflearn_amd import ZPs
    from mtflearn_amd.features import pca
    from mtflearn_amd.synthetic import honeycomb_frame, sliding_patches
    frame = honeycomb_frame(256, seed=4)
    patches = sliding_patches(frame, 32, rows=range(0, 224, 2), cols=range(0, 224, 2))       # 12 544 patches
    X = ZPs(8, 32).transform(patches).data
    for k in (2, 5, 20, 45):
        model = PCA(n_components=k)
        ref = model.fit_transform(X)
        assert model._fit_svd_solver == "covariance_eigh"
        got = pca(X, n_components=k)
        assert got.shape == ref.shape and got.dtype == np.float64
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    rng = np.random.default_rng(0)                                                            # other widths, float32 input
    for n, d in ((5000, 3), (4097, 66), (2000, 127)):
        Y = (rng.standard_normal((n, d)) * rng.random(d) * 3 + rng.random(d)).astype(np.float32)
        ref = PCA(n_components=2).fit_transform(Y.astype(np.float64))
        np.testing.assert_allclose(pca(Y, 2), ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    with pytest.raises(ValueError, match="must be between 1 and"):
        pca(X, n_components=46)
    with pytest.raises(ValueError, match="Expected 2D array"):
        pca(np.zeros(5))
