"""Downstream consumers of the moment matrix.

``pca(X, n_components)`` replaces ``mtflearn.features.pca`` (reference ``mtflearn/features/_dimension_reduction.py:3-6``:
``sklearn.decomposition.PCA(n_components).fit_transform(X)``) with the two passes over ``X`` on the GPU -- the Gram
matrix with the column sums (``zk_gram``) and the projection (``zk_project``), ``csrc/zk_consumers.hip`` -- and the
``D x D`` eigen-problem between them on the host, following the arithmetic of scikit-learn's ``covariance_eigh`` solver
(``sklearn/decomposition/_pca.py``: ``C = (X^T X - n mean mean^T) / (n - 1)``, ``eigh``, descending order, negative
eigenvalues clipped, ``svd_flip(u_based_decision=False)``: the entry of largest magnitude of every component is positive).
scikit-learn is a dependency of the reference and of this package, so parity is tested against it directly
(``tests/test_gpu_consumers.py``).  scikit-learn picks that solver itself when ``n_samples >= 10 n_features`` (the
usual case: millions of patches x 45 moments); for smaller inputs it uses an SVD of the centred matrix, which this
routine does not imitate -- same subspace and signs, last digits may differ.

The reference's clustering wrappers (``kmeans_lbs`` / ``gmm_lbs``) live in ``mtflearn_amd.clustering``, ``ForceGraph8`` in
``mtflearn_amd.manifold``.
"""
from __future__ import annotations

from ctypes import POINTER, byref, c_double, c_void_p

import numpy as np

from .. import _native
from .pickers import _device

__all__ = ["pca"]


def _covariance_eigh(gram, n_rows):
    """(mean, components (D, D) rows sorted by decreasing variance, explained_variance) from G = [X|1]^T [X|1]."""
    d = gram.shape[0] - 1
    mean = gram[d, :d] / n_rows
    cov = gram[:d, :d] - n_rows * np.outer(mean, mean)
    cov /= n_rows - 1
    from ..clustering import _one_blas_thread
    with _one_blas_thread():                                   # a D x D problem: threaded LAPACK only pays for waking its workers
        eigenvals, eigenvecs = np.linalg.eigh(cov)
    eigenvals = np.flip(eigenvals, axis=0).copy()
    eigenvecs = np.flip(eigenvecs, axis=1)
    eigenvals[eigenvals < 0.0] = 0.0
    vt = eigenvecs.T.copy()
    idx = np.argmax(np.abs(vt), axis=1)                                     # svd_flip(u_based_decision=False)
    signs = np.sign(vt[np.arange(vt.shape[0]), idx])
    vt *= signs[:, None]
    return mean, vt, eigenvals


def _pca_resident(rows, n_components, comm):
    """``pca`` of a matrix that is already on the device (``clustering.DeviceRows``), or of one spread over the ranks of a
    communicator: the Gram matrix about the column means (``zk_rows_gram``, summed over the ranks in rank order), ``eigh``,
    and the projection of this rank's rows.  Same ``covariance_eigh`` arithmetic as below up to the centring of the sums."""
    from ..clustering import _Shards
    sh = _Shards(rows, comm)
    n, d = sh.total, rows.n_features
    k = min(n, d) if n_components is None else n_components
    if not isinstance(k, (int, np.integer)) or not 1 <= k <= min(n, d):
        raise ValueError(f"n_components={n_components!r} must be between 1 and min(n_samples, n_features)={min(n, d)}")
    if n < 2:
        raise ValueError("PCA needs at least two samples")
    shift = sh.sum(rows.colsum()) / n
    mean, vt, _ = _covariance_eigh(sh.sum(rows.gram(shift)), n)
    return rows.project(mean + shift, vt[:k])


def pca(X, n_components=2, reconstruct=False, comm=None):
    """First ``n_components`` principal-component scores of ``X`` (N, D) -> (N, n_components) float64.

    ``X`` may be a ``mtflearn_amd.clustering.DeviceRows`` (a matrix already resident on the GPU); with ``comm`` it is this
    rank's block of rows of a matrix spread over the ranks and the components are those of the whole matrix (the ranks
    exchange one ``(D + 1) x (D + 1)`` Gram matrix); the scores of this rank's rows are returned."""
    from ..clustering import DeviceRows
    if isinstance(X, DeviceRows) or (hasattr(X, "gram") and hasattr(X, "project")):   # (tests plug a NumPy stand-in here)
        return _pca_resident(X, n_components, comm)
    if comm is not None and comm.world > 1:
        from ..clustering import _as_rows
        rows, _ = _as_rows(X)
        try:
            return _pca_resident(rows, n_components, comm)
        finally:
            rows.close()
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError(f"Expected 2D array, got {X.ndim}D array instead")
    n, d = X.shape
    k = min(n, d) if n_components is None else n_components
    if not isinstance(k, (int, np.integer)) or not 1 <= k <= min(n, d):
        raise ValueError(f"n_components={n_components!r} must be between 1 and min(n_samples, n_features)={min(n, d)}")
    if n < 2:
        raise ValueError("PCA needs at least two samples")
    lib = _native.load()
    if _native.device_count() == 0:
        raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
    dev = _device()
    gram = np.empty((d + 1, d + 1), dtype=np.float64)
    x_dev = c_void_p()
    _native.check(lib.zk_gram(dev, X.ctypes.data_as(POINTER(c_double)), n, d, gram.ctypes.data_as(POINTER(c_double)),
                              byref(x_dev)), "zk_gram")
    freed = False                                                            # the last zk_project call frees x_dev
    try:
        mean, vt, _ = _covariance_eigh(gram, n)
        out = _native.pinned.empty((n, k))
        done = 0
        while done < k:                                                      # at most 16 components per launch
            kk = min(16, k - done)
            comp = np.ascontiguousarray(vt[done:done + kk])
            part = out if kk == k else np.empty((n, kk), dtype=np.float64)
            last = done + kk == k
            freed = last                                                     # zk_project frees it on every exit path when asked to
            _native.check(lib.zk_project(dev, x_dev, n, d, mean.ctypes.data_as(POINTER(c_double)),
                                         comp.ctypes.data_as(POINTER(c_double)), kk, part.ctypes.data_as(POINTER(c_double)),
                                         int(last)), "zk_project")
            if part is not out:
                out[:, done:done + kk] = part
            done += kk
        return out
    finally:
        if not freed and x_dev.value:                                        # eigh / allocation / an earlier chunk failed
            lib.zk_device_free(dev, x_dev)
