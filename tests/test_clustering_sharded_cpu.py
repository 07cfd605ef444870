"""Sharded k-means / Gaussian mixture on CPU: world_size-2 and -3 gloo processes run the product's control flow (``mtflearn_amd.clustering.
kmeans_fit`` / ``kmeans_lbs`` / ``gmm_fit_predict`` / ``gmm_lbs`` with ``comm=``) with the test-aid communicator ``TorchComm`` and a NumPy stand-in for the
device passes (same methods as ``DeviceRows``, scikit-learn's formulas in NumPy).  Under test is everything around the kernels:
the rank-ordered sums, the owner look-up of the seeding draws over block boundaries (ragged and empty blocks), fetching
candidate rows from their owners, the label gather.  The result must be scikit-learn's clustering of the WHOLE matrix, and
identical to the single-block run of the same code.  (The kernels themselves: tests/test_gpu_consumers.py.)"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


class HostRows:
    """NumPy stand-in for ``mtflearn_amd.clustering.DeviceRows`` (the passes of csrc/zk_cluster.hip)."""

    def __init__(self, X):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.n_rows, self.n_features = self.X.shape
        self.mean = np.zeros(self.n_features)
        self._labels = np.full(self.n_rows, -1, dtype=np.int32)
        self.closest = None
        self._cand = None

    def colsum(self):
        return self.X.sum(axis=0)

    def center_at(self, mean):
        self.mean = np.array(mean)
        self.Xc = self.X - self.mean
        self.xsq = np.einsum("ij,ij->i", self.Xc, self.Xc)
        return (self.Xc ** 2).sum(axis=0), int(np.sum(~np.isfinite(self.xsq)))

    def fetch(self, idx, centred=True):
        return self.Xc[np.asarray(idx, dtype=np.int64)] if centred else self.X[np.asarray(idx, dtype=np.int64)]

    def seed_step(self, cand, cand_sq, use_closest):
        d = np.maximum((-2.0 * cand @ self.Xc.T + np.asarray(cand_sq)[:, None]) + self.xsq[None, :], 0.0)
        if use_closest:
            d = np.minimum(d, self.closest[None, :])
        self._cand = d
        return d.sum(axis=1)

    def seed_pick(self, which, vals=()):
        self.closest = self._cand[which]
        vals = np.asarray(vals, dtype=np.float64)
        if self.n_rows == 0:
            return np.zeros(len(vals), dtype=np.int64)
        return np.minimum(np.searchsorted(np.cumsum(self.closest), vals), self.n_rows - 1).astype(np.int64)

    def reset_labels(self):
        self._labels[:] = -1

    def lloyd(self, centers, update=True):
        k = len(centers)
        score = (centers ** 2).sum(axis=1)[None, :] - 2.0 * self.Xc @ centers.T
        new = np.argmin(score, axis=1).astype(np.int32) if self.n_rows else np.zeros(0, dtype=np.int32)
        changed = int(np.sum(new != self._labels))
        self._labels = new
        sums, counts = np.zeros((k, self.n_features)), np.zeros(k)
        if update:
            np.add.at(sums, new, self.Xc)
            counts = np.bincount(new, minlength=k).astype(np.float64)
        return sums, counts, changed

    def own_distance(self, centers):
        return ((self.Xc - centers[self._labels]) ** 2).sum(axis=1)

    def labels(self):
        return self._labels.copy()

    # ---- mixture passes (sklearn/mixture/_gaussian_mixture.py formulas) ----
    def resp_from_labels(self, k):
        self.resp = np.zeros((self.n_rows, k))
        self.resp[np.arange(self.n_rows), self._labels] = 1.0

    def estep(self, prec_chol, means, log_det, log_w, want_resp=True):
        from scipy.special import logsumexp
        k, d = means.shape
        lp = np.empty((self.n_rows, k))
        for c in range(k):
            y = self.X @ prec_chol[c] - means[c] @ prec_chol[c]
            lp[:, c] = (-0.5 * (d * np.log(2 * np.pi) + np.sum(y * y, axis=1)) + log_det[c]) + log_w[c]
        lse = logsumexp(lp, axis=1) if self.n_rows else np.zeros(0)
        if want_resp:
            self.resp = np.exp(lp - lse[:, None])
        self._labels = lp.argmax(axis=1).astype(np.int32) if self.n_rows else np.zeros(0, dtype=np.int32)
        return float(lse.sum())

    def gram(self, shift):
        z = np.concatenate([self.X - shift, np.ones((self.n_rows, 1))], axis=1)
        return z.T @ z

    def project(self, mean, components):
        return (self.X - mean) @ np.asarray(components).T

    def moments(self, component, shift, count=None):
        z = np.concatenate([self.X - shift, np.ones((self.n_rows, 1))], axis=1)
        one = lambda c: (z * self.resp[:, c][:, None]).T @ z
        return one(component) if count is None else np.stack([one(component + c) for c in range(count)])

    def close(self):
        pass


def _data():
    rng = np.random.default_rng(11)
    centres = rng.standard_normal((5, 9)) * 2.5
    return centres[rng.integers(0, 5, 1500)] + rng.standard_normal((1500, 9))


def _blocks(n, world, kind):
    if kind == "even":
        cuts = np.linspace(0, n, world + 1).astype(int)
    elif kind == "ragged":
        cuts = np.concatenate([[0], np.sort(np.random.default_rng(world).choice(np.arange(1, n), world - 1, replace=False)), [n]])
    else:                                                # an empty block in the middle / at the end
        cuts = np.array([0, n // 3, n // 3, n][:world + 1] if world == 3 else [0, n, n])
    return cuts


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    for p in (ROOT, os.path.join(ROOT, "motif-learn_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from mtflearn_amd import distributed as D
    from mtflearn_amd import clustering as C
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = D.TorchComm()
        X = _data()
        C._as_rows = lambda rows: (rows, False)                      # the stand-in goes where a DeviceRows would
        for kind in ("even", "ragged", "empty"):
            cuts = _blocks(len(X), world, kind)
            for k, seed in ((5, 0), (3, 4), (8, 1)):
                rows = HostRows(X[cuts[rank]:cuts[rank + 1]])
                labels, centers, n_iter = C.kmeans_fit(rows, k, random_state=seed, comm=comm)
                assert labels.shape == (cuts[rank + 1] - cuts[rank],)
                whole = C.gather_labels(labels, comm)
                lbs = C.kmeans_lbs(HostRows(X[cuts[rank]:cuts[rank + 1]]), k, random_state=seed, comm=comm)
                np.savez(os.path.join(tmpdir, f"{kind}_{k}_{seed}_{rank}.npz"), labels=whole, centers=centers, n_iter=n_iter, lbs=lbs)
            from mtflearn_amd.features import pca
            scores = pca(HostRows(X[cuts[rank]:cuts[rank + 1]]), 3, comm=comm)
            np.save(os.path.join(tmpdir, f"pca_{kind}_{rank}.npy"), scores)
            for k, cov in ((3, "full"), (5, "diag"), (4, "tied"), (5, "spherical")):
                mine, n_iter, conv = C.gmm_fit_predict(HostRows(X[cuts[rank]:cuts[rank + 1]]), k, covariance_type=cov, comm=comm)
                lbs = C.gmm_lbs(HostRows(X[cuts[rank]:cuts[rank + 1]]), k, type=cov, comm=comm)
                np.savez(os.path.join(tmpdir, f"gmm_{kind}_{k}_{cov}_{rank}.npz"), labels=C.gather_labels(mine, comm), n_iter=n_iter,
                         converged=conv, lbs=lbs)
        comm.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_clustering_is_the_clustering_of_the_whole_matrix(tmp_path, world):
    import torch.multiprocessing as mp
    from sklearn.cluster import KMeans
    from mtflearn_amd import clustering as C
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    X = _data()
    saved = C._as_rows
    C._as_rows = lambda rows: (rows, False)
    try:
        for k, seed in ((5, 0), (3, 4), (8, 1)):
            model = KMeans(n_clusters=k, random_state=seed).fit(X)
            one_labels, one_centers, one_iter = C.kmeans_fit(HostRows(X), k, random_state=seed)     # the same code, one block
            np.testing.assert_array_equal(one_labels, model.labels_)
            assert one_iter == model.n_iter_
            for kind in ("even", "ragged", "empty"):
                for rank in range(world):
                    with np.load(tmp_path / f"{kind}_{k}_{seed}_{rank}.npz") as f:
                        np.testing.assert_array_equal(f["labels"], model.labels_, err_msg=f"{kind} {k} {seed} rank {rank}")
                        np.testing.assert_allclose(f["centers"], model.cluster_centers_, rtol=0, atol=1e-9)
                        assert int(f["n_iter"]) == model.n_iter_
                        np.testing.assert_array_equal(f["lbs"], C._relabel_by_size(model.labels_))
        from sklearn.decomposition import PCA
        ref_scores = PCA(n_components=3).fit_transform(X)
        for kind in ("even", "ragged", "empty"):
            cuts = _blocks(len(X), world, kind)
            got = np.concatenate([np.load(tmp_path / f"pca_{kind}_{rank}.npy") for rank in range(world)])
            np.testing.assert_allclose(got, ref_scores, rtol=0, atol=1e-10 * np.abs(ref_scores).max())
        from sklearn.mixture import GaussianMixture
        for k, cov in ((3, "full"), (5, "diag"), (4, "tied"), (5, "spherical")):
            model = GaussianMixture(k, covariance_type=cov, random_state=0).fit(X)
            ref = model.predict(X)
            for kind in ("even", "ragged", "empty"):
                for rank in range(world):
                    with np.load(tmp_path / f"gmm_{kind}_{k}_{cov}_{rank}.npz") as f:
                        np.testing.assert_array_equal(f["labels"], ref, err_msg=f"gmm {kind} {k} {cov} rank {rank}")
                        assert int(f["n_iter"]) == model.n_iter_ and bool(f["converged"]) == model.converged_
                        np.testing.assert_array_equal(f["lbs"], C._relabel_by_size(ref.astype(np.intp)))
    finally:
        C._as_rows = saved


def test_empty_cluster_relocation_single_and_sharded():
    """``_relocate_empty_clusters``: one block reproduces scikit-learn's routine; several blocks pick the globally farthest
    points (here checked on one block through the same code path by a one-rank stand-in communicator)."""
    from mtflearn_amd import clustering as C

    class OneRank:
        rank, world = 0, 2                                # world 2 selects the sharded branch

        def allgather_host(self, payload):
            return [payload, bytes(len(payload))]        # the other rank: no rows, zero sums, offers at distance 0

    rng = np.random.default_rng(3)
    X = rng.standard_normal((200, 4))
    centers_old = np.array([[0.0] * 4, [0.1] * 4, [50.0] * 4])          # nobody is nearest to the third centre
    for comm in (None, OneRank()):
        rows = HostRows(X)
        sh = C._Shards(rows, comm)
        sh.center()
        sums, counts, _ = rows.lloyd(centers_old - rows.mean, update=True)
        assert counts[2] == 0
        before = sums.copy()
        C._relocate_empty_clusters(sh, centers_old - rows.mean, sums, counts)
        far = int(np.argmax(rows.own_distance(centers_old - rows.mean)))
        assert counts[2] == 1 and counts.sum() == 200
        np.testing.assert_allclose(sums[2], rows.Xc[far])
        np.testing.assert_allclose(sums.sum(axis=0), before.sum(axis=0), rtol=0, atol=1e-12)
