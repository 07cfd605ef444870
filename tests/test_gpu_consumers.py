"""pca on the device against scikit-learn's PCA, which is what the reference's pca() calls."""
import os
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


@pytest.fixture(scope="module")
def consumers_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumers_golden.npz"))


def test_clustering_labels_match_the_reference(consumers_golden):
    """kmeans_lbs / gmm_lbs on the device against labels captured from the reference's own two functions
    (clustering/_clustering_functions.py:8-33, oracle/make_golden_consumers.py)."""
    from mtflearn_amd.clustering import kmeans_lbs, gmm_lbs
    g = consumers_golden
    for key in g.files:
        parts = key.split("_")
        if parts[0] == "kmeans":
            X, n = g[parts[1]], int(parts[2])
            got = kmeans_lbs(X, n, random_state=7) if key.endswith("rs7") else kmeans_lbs(X, n)
        elif parts[0] == "gmm":
            X, n = g[parts[1]], int(parts[2])
            got = gmm_lbs(X, n, type=parts[3])
        else:
            continue
        assert got.dtype == g[key].dtype and got.shape == g[key].shape, key
        np.testing.assert_array_equal(got, g[key], err_msg=key)


def _moment_matrix(n_max=8, step=2, noise=True):
    from mtflearn_amd import ZPs
    from mtflearn_amd.synthetic import honeycomb_frame, sliding_patches
    frame = honeycomb_frame(320, seed=11, noise=noise)
    patches = sliding_patches(frame, 32, rows=range(0, 288, step), cols=range(0, 288, step))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ZPs(n_max, 32).transform(patches).data


def test_kmeans_matches_sklearn_on_moment_matrices():
    """Overlapping clusters (tens of Lloyd iterations), several k and seeds, 45 and 91 features: labels and centres against
    scikit-learn's KMeans itself.  A label may differ only where two centres are equidistant to rounding."""
    from sklearn.cluster import KMeans
    from mtflearn_amd.clustering import kmeans_fit, DeviceRows
    X = _moment_matrix()
    rng = np.random.default_rng(1)
    cases = [(X, 3, 0), (X, 6, 1), (X, 9, 5), (_moment_matrix(12, 3, False), 4, 0),
             (rng.standard_normal((5000, 3)), 8, 2), (rng.standard_normal((777, 127)) + 3.0, 5, 0),
             (rng.standard_normal((300, 64)), 17, 3), (rng.standard_normal((64, 2)), 1, 0)]
    for data, k, seed in cases:
        model = KMeans(n_clusters=k, random_state=seed).fit(data)
        labels, centers, n_iter = kmeans_fit(data, k, random_state=seed)
        agree = np.mean(labels == model.labels_)
        assert agree >= 0.999, (data.shape, k, seed, agree, n_iter, model.n_iter_)
        np.testing.assert_allclose(centers, model.cluster_centers_, rtol=0, atol=1e-6 * np.abs(model.cluster_centers_).max())
        assert n_iter == model.n_iter_
    # a resident matrix is clustered repeatedly without another upload; float32 input is promoted
    with DeviceRows(X.astype(np.float32)) as rows:
        a = kmeans_fit(rows, 4, random_state=0)[0]
        b = kmeans_fit(rows, 4, random_state=0)[0]
        np.testing.assert_array_equal(a, b)
    ref = KMeans(n_clusters=4, random_state=0).fit(X.astype(np.float32).astype(np.float64)).labels_
    assert np.mean(a == ref) >= 0.999
    with pytest.raises(ValueError, match="should be >= n_clusters"):
        kmeans_fit(X[:3], 5)
    bad = X[:100].copy()
    bad[7, 3] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        kmeans_fit(bad, 2)


def test_gmm_matches_sklearn_on_moment_matrices():
    from sklearn.mixture import GaussianMixture
    from mtflearn_amd.clustering import gmm_fit_predict
    X = _moment_matrix()
    pcs = X @ np.linalg.svd(X - X.mean(0), full_matrices=False)[2][:12].T            # 12 leading components: well-conditioned
    rng = np.random.default_rng(3)
    blobs = np.concatenate([rng.standard_normal((3000, 20)) * s + c for s, c in ((1.0, 0.0), (0.5, 3.0), (2.0, -4.0))])
    for data, k, kind in [(pcs, 3, "full"), (pcs, 4, "diag"), (pcs, 3, "tied"), (pcs, 3, "spherical"), (blobs, 3, "full"),
                          (blobs, 5, "full"), (X, 2, "full")]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = GaussianMixture(k, covariance_type=kind, random_state=0).fit(data)
            ref = model.predict(data)
            labels, n_iter, converged = gmm_fit_predict(data, k, covariance_type=kind, random_state=0)
        assert np.mean(labels == ref) >= 0.999, (data.shape, k, kind, np.mean(labels == ref), n_iter, model.n_iter_)
        assert n_iter == model.n_iter_ and converged == model.converged_


def test_kmeans_on_the_matrix_where_the_batch_kernel_left_it():
    """The device-resident flow: patches -> zk_transform_patches_dev -> DeviceRows.adopt -> labels; the moments never visit
    the host before the comparison copy.  Also through a world-1 RCCL communicator (the sharded code path's entry)."""
    import torch
    from sklearn.cluster import KMeans
    from mtflearn_amd import ZPs, distributed as D
    from mtflearn_amd.clustering import DeviceRows, kmeans_fit, kmeans_lbs, _relabel_by_size
    from mtflearn_amd.synthetic import honeycomb_frame
    frame = torch.from_numpy(honeycomb_frame(256, seed=2)).cuda()
    patches = frame.unfold(0, 32, 2).unfold(1, 32, 2).reshape(-1, 32, 32).contiguous()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plan = ZPs(8, 32)._device_plan()
    moments = D.patch_moments_device(plan, patches)
    torch.cuda.synchronize()
    ref = KMeans(n_clusters=5, random_state=0).fit(moments.cpu().numpy())
    with DeviceRows.adopt(moments.data_ptr(), moments.shape[0], moments.shape[1], device=0) as rows:
        labels, centers, n_iter = kmeans_fit(rows, 5, random_state=0)
        assert np.mean(labels == ref.labels_) >= 0.999 and n_iter == ref.n_iter_
        comm = D.RcclComm(0, 0, 1, path=f"/tmp/zk_test_cluster_{os.getpid()}.id")
        try:
            np.testing.assert_array_equal(kmeans_fit(rows, 5, random_state=0, comm=comm)[0], labels)
            np.testing.assert_array_equal(kmeans_lbs(rows, 5, comm=comm), _relabel_by_size(labels))
        finally:
            comm.close()
    assert torch.isfinite(moments).all()                         # an adopted matrix is borrowed, not freed


def test_pca_of_a_resident_matrix():
    """pca(DeviceRows): Gram matrix in a fixed order (zk_rows_gram) + projection, the matrix staying where it is."""
    from sklearn.decomposition import PCA
    from mtflearn_amd.features import pca
    from mtflearn_amd.clustering import DeviceRows, kmeans_fit
    X = _moment_matrix()
    with DeviceRows(X) as rows:
        for k in (2, 20, 45):
            ref = PCA(n_components=k).fit_transform(X)
            got = pca(rows, n_components=k)
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
        np.testing.assert_array_equal(pca(rows, 3), pca(rows, 3))              # bit-identical from run to run
        kmeans_fit(rows, 3)                                                     # and the same resident matrix clusters on


def test_row_passes_on_edge_shapes():
    """Every pass of csrc/zk_cluster.hip against its NumPy statement (the stand-in of tests/test_clustering_sharded_cpu.py) on
    the shapes where tiling can go wrong: 1 to 127 features (odd, even, exactly 64), row counts below / at / just past a
    64-row tile, one cluster to more than the register-resident limit, ragged last tiles."""
    from scipy import linalg
    from test_clustering_sharded_cpu import HostRows
    from mtflearn_amd.clustering import DeviceRows
    rng = np.random.default_rng(99)
    shapes = [(1, 1), (2, 1), (63, 2), (64, 3), (65, 64), (129, 127), (1000, 45), (4097, 66), (200, 91), (777, 5), (5000, 120),
              (64 * 7, 33), (64 * 7 + 1, 80), (64 * 7 - 1, 81)]
    for n, d in shapes:
        X = rng.standard_normal((n, d)) * (1 + rng.random(d)) + rng.standard_normal(d) * 3
        host = HostRows(X)
        with DeviceRows(X) as dev:
            np.testing.assert_allclose(dev.colsum(), host.colsum(), rtol=1e-12, atol=1e-9)
            mean = host.colsum() / n
            sq_d, bad_d = dev.center_at(mean)
            sq_h, bad_h = host.center_at(mean)
            np.testing.assert_allclose(sq_d, sq_h, rtol=1e-12, atol=1e-12)
            assert bad_d == bad_h == 0
            idx = rng.integers(0, n, min(n, 5))
            np.testing.assert_array_equal(dev.fetch(idx), host.fetch(idx))
            for t in (1, 4, 7):
                cand = host.fetch(rng.integers(0, n, t))
                csq = np.einsum("ij,ij->i", cand, cand)
                np.testing.assert_allclose(dev.seed_step(cand, csq, False), host.seed_step(cand, csq, False), rtol=1e-11, atol=1e-9)
                which = int(rng.integers(0, t))
                vals = np.sort(rng.random(3)) * host._cand[which].sum()
                pick_d, pick_h = dev.seed_pick(which, vals), host.seed_pick(which, vals)
                near = np.abs(np.cumsum(host.closest)[pick_h] - vals) < 1e-9 * max(1.0, host.closest.sum())
                assert np.all((pick_d == pick_h) | near)
                np.testing.assert_allclose(dev.seed_step(cand, csq, True), host.seed_step(cand, csq, True), rtol=1e-11, atol=1e-9)
            for k in (1, 3, 8, 13, 16, 17, 40):
                if k > n:
                    continue
                centers = host.fetch(rng.choice(n, k, replace=False)) + 0.01 * rng.standard_normal((k, d))
                dev.reset_labels(), host.reset_labels()
                for update in (True, False):
                    s_d, c_d, ch_d = dev.lloyd(centers, update)
                    s_h, c_h, ch_h = host.lloyd(centers, update)
                    np.testing.assert_array_equal(dev.labels(), host.labels())
                    assert ch_d == ch_h
                    if update:
                        np.testing.assert_array_equal(c_d, c_h)
                        np.testing.assert_allclose(s_d, s_h, rtol=1e-11, atol=1e-10)
                np.testing.assert_allclose(dev.own_distance(centers), host.own_distance(centers), rtol=1e-11, atol=1e-12)
            for k in (1, 2, 5, 12):
                if k > n or d > 91 and k > 5:
                    continue
                means = X[rng.choice(n, k, replace=False)]
                prec = np.zeros((k, d, d))
                for c in range(k):
                    a = rng.standard_normal((d, d)) * 0.3 + np.eye(d) * 2
                    prec[c] = linalg.solve_triangular(linalg.cholesky(a @ a.T, lower=True), np.eye(d), lower=True).T
                log_det = np.log(np.einsum("kii->ki", prec)).sum(axis=1)
                log_w = np.log(np.full(k, 1.0 / k))
                lse_d, lse_h = dev.estep(prec, means, log_det, log_w), host.estep(prec, means, log_det, log_w)
                assert abs(lse_d - lse_h) <= 1e-10 * max(1.0, abs(lse_h))
                agree = np.mean(dev.labels() == host.labels())
                assert agree == 1.0 or (agree > 0.99 and n > 500)
                shift = X.mean(axis=0)
                for c in range(k):
                    g_h = host.moments(c, shift)
                    np.testing.assert_allclose(dev.moments(c, shift), g_h, rtol=1e-10, atol=1e-10 * np.abs(g_h).max())
                dev.lloyd(means - mean, False), host.lloyd(means - mean, False)
                dev.resp_from_labels(k), host.resp_from_labels(k)
                np.testing.assert_allclose(dev.moments(0, shift), host.moments(0, shift), rtol=1e-11, atol=1e-9)
            g_h = host.gram(mean)
            np.testing.assert_allclose(dev.gram(mean), g_h, rtol=1e-11, atol=1e-11 * np.abs(g_h).max())
            comp = rng.standard_normal((min(d, 20), d))
            np.testing.assert_allclose(dev.project(mean, comp), host.project(mean, comp), rtol=1e-11, atol=1e-11)


def test_mixture_e_step_on_the_matrix_cores():
    """zk_gmm_estep for D <= 48 and k <= 8 runs as y = x P on the matrix cores (estep_mfma_kernel): the whole upper triangle for
    'full' / 'tied' factors, the diagonal pieces only for 'diag' / 'spherical' ones; 1 to 3 column blocks, row counts around
    the 16-row blocks of a wave and the 128 rows of a workgroup round; against the NumPy statement -- log-likelihood, labels and
    (through the weighted moments) the responsibilities."""
    from scipy import linalg
    from test_clustering_sharded_cpu import HostRows
    from mtflearn_amd.clustering import DeviceRows
    rng = np.random.default_rng(5)
    for n, d in [(1, 2), (15, 16), (16, 17), (17, 32), (127, 33), (128, 45), (129, 48), (1000, 45), (4099, 40), (70001, 45)]:
        X = rng.standard_normal((n, d)) * (1 + rng.random(d)) + rng.standard_normal(d) * 2
        host = HostRows(X)
        with DeviceRows(X) as dev:
            for k in (1, 4, 8):
                if k > n:
                    continue
                for kind in ("full", "diag"):
                    means = X[rng.choice(n, k, replace=False)] + 0.1 * rng.standard_normal((k, d))
                    prec = np.zeros((k, d, d))
                    for c in range(k):
                        if kind == "full":
                            a = rng.standard_normal((d, d)) * 0.3 + np.eye(d) * 2
                            prec[c] = linalg.solve_triangular(linalg.cholesky(a @ a.T, lower=True), np.eye(d), lower=True).T
                        else:
                            prec[c] = np.diag(0.5 + rng.random(d))
                    log_det = np.log(np.einsum("kii->ki", prec)).sum(axis=1)
                    log_w = np.log(rng.dirichlet(np.ones(k) * 5))
                    lse_d, lse_h = dev.estep(prec, means, log_det, log_w), host.estep(prec, means, log_det, log_w)
                    assert abs(lse_d - lse_h) <= 1e-10 * max(1.0, abs(lse_h)), (n, d, k, kind)
                    agree = np.mean(dev.labels() == host.labels())
                    assert agree == 1.0 or (agree > 0.999 and n > 500), (n, d, k, kind, agree)
                    shift = X.mean(axis=0)
                    for c in range(k):
                        g_h = host.moments(c, shift)
                        np.testing.assert_allclose(dev.moments(c, shift), g_h, rtol=1e-10, atol=1e-10 * np.abs(g_h).max())


def test_sharded_control_flow_on_the_device_passes():
    """The sharded code path (``comm=``) driving real device passes: a two-rank communicator whose other rank holds no rows and
    contributes zeros to every sum -- labels, centres, mixture labels and PCA scores must equal the single-block run."""
    from mtflearn_amd.clustering import DeviceRows, kmeans_fit, gmm_fit_predict, kmeans_lbs, _relabel_by_size
    from mtflearn_amd.features import pca

    class OtherRankEmpty:
        rank, world = 0, 2

        def allgather_host(self, payload):
            return [payload, bytes(len(payload))]

    X = _moment_matrix(step=3)
    comm = OtherRankEmpty()
    with DeviceRows(X) as rows:
        one = kmeans_fit(rows, 5, random_state=2)
        two = kmeans_fit(rows, 5, random_state=2, comm=comm)
        np.testing.assert_array_equal(one[0], two[0])
        np.testing.assert_array_equal(one[1], two[1])
        assert one[2] == two[2]
        np.testing.assert_array_equal(kmeans_lbs(rows, 5, random_state=2, comm=comm), _relabel_by_size(one[0]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g1 = gmm_fit_predict(rows, 3, covariance_type="full", random_state=1)
            g2 = gmm_fit_predict(rows, 3, covariance_type="full", random_state=1, comm=comm)
        np.testing.assert_array_equal(g1[0], g2[0])
        assert g1[1:] == g2[1:]
        np.testing.assert_array_equal(pca(rows, 4), pca(rows, 4, comm=comm))
