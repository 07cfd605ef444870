"""The manifold consumer on the device: the neighbour search and affinities of ForceGraph8's
``compute_graph`` against scikit-learn's ``NearestNeighbors(metric='correlation')`` (what the reference calls) and the oracle's
restatement of ``calculate_asymmetric_Pij`` / ``calculate_graph``; the class end to end against the oracle pipeline."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mo():
    from oracle import manifold_oracle
    return manifold_oracle


def _moments(n_max=8, step=5):
    from mtflearn_amd import ZPs
    from mtflearn_amd.synthetic import honeycomb_frame, sliding_patches
    frame = honeycomb_frame(320, seed=21)
    patches = sliding_patches(frame, 32, rows=range(0, 288, step), cols=range(0, 288, step))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ZPs(n_max, 32).transform(patches).data


def test_neighbours_and_affinities(mo):
    from sklearn.neighbors import NearestNeighbors
    from mtflearn_amd.clustering import DeviceRows
    from mtflearn_amd.manifold import _knn_affinities, compute_graph
    rng = np.random.default_rng(2)
    for X, k in [(_moments(), 10), (rng.standard_normal((1000, 7)), 5), (rng.standard_normal((777, 91)) + 2.0, 17),
                 (rng.standard_normal((130, 45)), 40), (rng.standard_normal((64, 3)), 1)]:
        d_ref, i_ref = NearestNeighbors(algorithm="auto", n_neighbors=k, metric="correlation").fit(X).kneighbors(X, n_neighbors=k)
        with DeviceRows(X) as rows:
            lc = min(1, k - 1)
            dist, ind, P = _knn_affinities(rows, k, lc, k)
        np.testing.assert_array_equal(ind, i_ref)
        np.testing.assert_allclose(dist, d_ref, rtol=0, atol=1e-13)
        if k > 1:
            np.testing.assert_allclose(P, mo.calculate_asymmetric_Pij(d_ref, perplexity=k, local_conectivity=lc), rtol=1e-9, atol=1e-15)
            graph, nbrs = compute_graph(X, k, "correlation", None, lc, 1.0)
            g_ref, n_ref = mo.compute_graph(X, k, "correlation", None, lc, 1.0)
            np.testing.assert_array_equal(nbrs, n_ref)
            assert (graph != graph.T).nnz == 0
            np.testing.assert_allclose(graph.toarray(), g_ref.toarray(), rtol=1e-9, atol=1e-15)
    with pytest.raises(ValueError, match="correlation"):
        compute_graph(X, 3, "euclidean")


def test_matrix_core_search_against_the_scalar_kernel_and_sklearn(monkeypatch):
    """k <= 16, D <= 96 runs as a GEMM on the matrix cores (knn_mfma_kernel): ragged row counts around the 16 / 64 / 128-row
    blocking, 1-24 feature steps, exact ties (duplicated rows keep the smaller index first, as the scalar-operand kernel does)."""
    from sklearn.neighbors import NearestNeighbors
    from mtflearn_amd.clustering import DeviceRows
    from mtflearn_amd.manifold import _knn_affinities
    rng = np.random.default_rng(11)
    cases = [(16, 4, 16), (17, 45, 3), (63, 45, 10), (65, 45, 16), (129, 28, 3), (257, 5, 1), (777, 91, 16), (1000, 96, 12),
             (5000, 45, 16), (4097, 66, 10)]
    for n, d, k in cases:
        X = rng.standard_normal((n, d)) + rng.standard_normal(d)
        d_ref, i_ref = NearestNeighbors(n_neighbors=k, metric="correlation").fit(X).kneighbors(X, n_neighbors=k)
        with DeviceRows(X) as rows:
            dist, ind, _ = _knn_affinities(rows, k, min(1, k - 1), k)
        np.testing.assert_array_equal(ind, i_ref, err_msg=str((n, d, k)))
        np.testing.assert_allclose(dist, d_ref, rtol=0, atol=1e-13)
    # exact ties: every row three times, shuffled
    base = rng.standard_normal((700, 45))
    X = np.concatenate([base, base, base])[rng.permutation(2100)]
    with DeviceRows(X) as rows:
        d_m, i_m, _ = _knn_affinities(rows, 9, 1, 9)
        monkeypatch.setenv("ZK_KNN_SCALAR", "1")
        d_s, i_s, _ = _knn_affinities(rows, 9, 1, 9)
        monkeypatch.delenv("ZK_KNN_SCALAR")
        monkeypatch.setenv("ZK_KNN_PARTS", "2")          # the candidate range in two parts, merged by (distance, part)
        d_p, i_p, _ = _knn_affinities(rows, 9, 1, 9)
        monkeypatch.delenv("ZK_KNN_PARTS")
    np.testing.assert_array_equal(i_m, i_s)
    np.testing.assert_array_equal(i_p, i_s)
    np.testing.assert_array_equal(d_p, d_s)
    np.testing.assert_allclose(d_m, d_s, rtol=0, atol=1e-13)
    assert (np.sort(i_m[:, :3], axis=1) == np.sort(np.stack([np.flatnonzero((X == x).all(1)) for x in X]), axis=1)).all()


def test_force_graph8_end_to_end(mo):
    """Same graph (to rounding), same PCA start, same random state -> the sequential optimiser, which is bit-identical to the
    restatement given identical inputs (tests/test_manifold_cpu.py), lands on the same layout up to the amplification of the
    graph's last-digit differences over the sweeps."""
    from mtflearn_amd.manifold import ForceGraph8
    X = _moments(step=9)                                              # ~1 000 moment vectors
    for kw in (dict(num_iterations=10), dict(num_iterations=6, init_mode="random", random_state=3, n_neighbors=6, num_negative_samples=4)):
        fg = ForceGraph8(**kw)
        y = fg.fit_transform(X)
        y_ref, logs_ref, g_ref, nbrs_ref = mo.force_graph8(X, **kw)
        assert y.shape == (len(X), 2) and fg.nodes['x'].shape == (len(X),) and len(fg.logs) == len(logs_ref)
        np.testing.assert_array_equal(fg.nbrs_ind, nbrs_ref)
        np.testing.assert_allclose(fg.logs[0], logs_ref[0], rtol=0, atol=1e-9)                 # the initial layout
        np.testing.assert_allclose(fg.logs[2], logs_ref[2], rtol=0, atol=1e-6)                 # after the first sweep
        # the same optimiser on the product's own graph reproduces the product's layout exactly (host code path, from the GPU run)
        xy = np.ascontiguousarray(fg.logs[0]).copy()
        rs = np.random.RandomState(kw.get("random_state", 48))
        if kw.get("init_mode") == "random":
            rs.uniform(low=-10.0, high=10.0, size=(len(X), 2))
        state = [int(v) for v in rs.randint(mo.INT32_MIN, mo.INT32_MAX, 3).astype(np.int64)]
        mo.optimize_layout(kw["num_iterations"], xy, mo.compute_pairs(fg.graph), kw.get("num_negative_samples", 10), fg.nbrs_ind, 1.0,
                           np.array((0, 2, 1, 1)), np.array((2, 4, 5, 2)), state, 0.5)
        np.testing.assert_array_equal(xy, y)
        assert np.isfinite(y).all()
