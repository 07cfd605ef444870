"""Clustering consumers of the moment matrix on the GPU -- the mirror of ``mtflearn.clustering``.

``kmeans_lbs(X, n, random_state)`` and ``gmm_lbs(X, n, type, ramdom_state)`` replace the reference's wrappers
(``mtflearn/clustering/_clustering_functions.py:8-22, 25-33``): scikit-learn's ``KMeans(n, random_state).fit(X).labels_``
and ``GaussianMixture(n, covariance_type, random_state).fit(X).predict(X)``, each followed by the reference's relabelling by
cluster size.  The ``(N, D)`` matrix is uploaded once (or adopted where it already is: ``DeviceRows.adopt``) and every pass
over it -- squared distances to seeding candidates, Lloyd assignment with the per-cluster sums, the mixture's E step and the
weighted second moments of its M step -- is a kernel of ``csrc/zk_cluster.hip``.  What happens between passes is
scikit-learn's own control flow, restated here on ``k x D``-sized arrays: the k-means++ draws from the caller's
``RandomState`` (same calls in the same order, so the same candidates), centre averaging, the centre-shift / strict
convergence tests, empty-cluster relocation, Cholesky factors of the covariances (SciPy), the lower-bound test.  References
are to scikit-learn 1.7 (``sklearn/cluster/_kmeans.py``, ``_k_means_lloyd.pyx``, ``_k_means_common.pyx``,
``sklearn/mixture/_base.py``, ``_gaussian_mixture.py``), a dependency of the reference and of this package: parity is tested
against it directly, and against labels captured from the reference's own two functions (``tests/golden``).

Arithmetic differs from scikit-learn's only in summation order (its BLAS / OpenMP reductions are not reproducible from run
to run either); a label can differ where a point is equidistant from two centres to the last bits.  float32 input is
promoted to float64 (scikit-learn would cluster float32 data in single precision).
"""
from __future__ import annotations

import warnings
from ctypes import POINTER, byref, c_double, c_int32, c_int64, c_void_p

import os

import numpy as np

from . import _native
from .features.pickers import _device

__all__ = ["kmeans_lbs", "gmm_lbs", "sort_lbs", "seg_lbs", "normalize_xy", "DeviceRows", "kmeans_fit", "gmm_fit_predict",
           "gather_labels"]

_PD = POINTER(c_double)


def _p(a):
    return a.ctypes.data_as(_PD)


class DeviceRows:
    """A float64 matrix ``(N, D)``, ``D <= 127``, resident on one GPU with the work buffers of the clustering passes."""

    def __init__(self, X=None, device=None, _adopt=None):
        self._lib = _native.load()
        if _native.device_count() == 0:
            raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
        self.device = _device() if device is None else int(device)
        handle = c_void_p()
        if _adopt is not None:
            ptr, n, d = _adopt
            _native.check(self._lib.zk_rows_adopt(self.device, c_void_p(ptr), n, d, byref(handle)), "zk_rows_adopt")
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            if X.ndim != 2:
                raise ValueError(f"Expected 2D array, got {X.ndim}D array instead")
            n, d = X.shape
            if n == 0 or d == 0:
                raise ValueError(f"Found array with {n} sample(s) and {d} feature(s) while a minimum of 1 is required.")
            _native.check(self._lib.zk_rows_create(self.device, _p(X), n, d, byref(handle)), "zk_rows_create")
        self._h = handle
        self.n_rows, self.n_features = int(n), int(d)

    @classmethod
    def adopt(cls, device_pointer, n_rows, n_features, device=None):
        """Borrow a matrix that already lives in device memory (e.g. the output of ``zk_transform_patches_dev``)."""
        return cls(device=device, _adopt=(int(device_pointer), int(n_rows), int(n_features)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zk_rows_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- passes ----------------------------------------------------------------------------------------------------
    def colsum(self):
        out = np.empty(self.n_features)
        _native.check(self._lib.zk_rows_colsum(self._h, _p(out)), "zk_rows_colsum")
        return out

    def center_at(self, mean):
        """Make ``mean`` the centring shift; returns (sums of squared deviations per column, rows with a non-finite element)."""
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        sq, bad = np.empty(self.n_features), c_int64()
        _native.check(self._lib.zk_rows_center_at(self._h, _p(mean), _p(sq), byref(bad)), "zk_rows_center_at")
        return sq, int(bad.value)

    def fetch(self, idx, centred=True):
        idx = np.ascontiguousarray(idx, dtype=np.int64).ravel()
        out = np.empty((len(idx), self.n_features))
        _native.check(self._lib.zk_rows_fetch(self._h, idx.ctypes.data_as(POINTER(c_int64)), len(idx), int(centred), _p(out)),
                      "zk_rows_fetch")
        return out

    def seed_step(self, cand, cand_sq, use_closest):
        cand, cand_sq = np.ascontiguousarray(cand), np.ascontiguousarray(cand_sq)
        pot = np.empty(len(cand))
        _native.check(self._lib.zk_kmeans_seed_step(self._h, _p(cand), _p(cand_sq), len(cand), int(use_closest), _p(pot)),
                      "zk_kmeans_seed_step")
        return pot

    def seed_pick(self, which, vals=()):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        idx = np.empty(len(vals), dtype=np.int64)
        _native.check(self._lib.zk_kmeans_seed_pick(self._h, int(which), _p(vals), len(vals), idx.ctypes.data_as(POINTER(c_int64))),
                      "zk_kmeans_seed_pick")
        return idx

    def reset_labels(self):
        _native.check(self._lib.zk_rows_reset_labels(self._h), "zk_rows_reset_labels")

    def lloyd(self, centers, update=True):
        centers = np.ascontiguousarray(centers)
        k = len(centers)
        sums, counts, changed = np.empty((k, self.n_features)), np.empty(k), c_int64()
        _native.check(self._lib.zk_kmeans_step(self._h, _p(centers), k, int(update), _p(sums), _p(counts), byref(changed)),
                      "zk_kmeans_step")
        return sums, counts, int(changed.value)

    def profile(self, enable=True):
        _native.check(self._lib.zk_rows_profile(self._h, int(enable)), "zk_rows_profile")

    def last_kernel_ms(self):
        return float(self._lib.zk_rows_last_kernel_ms(self._h))

    def own_distance(self, centers):
        centers = np.ascontiguousarray(centers)
        out = np.empty(self.n_rows)
        _native.check(self._lib.zk_kmeans_own_distance(self._h, _p(centers), len(centers), _p(out)), "zk_kmeans_own_distance")
        return out

    def labels(self):
        out = np.empty(self.n_rows, dtype=np.int32)
        _native.check(self._lib.zk_rows_labels(self._h, out.ctypes.data_as(POINTER(c_int32))), "zk_rows_labels")
        return out

    def estep(self, prec_chol, means, log_det, log_w, want_resp=True):
        prec_chol, means = np.ascontiguousarray(prec_chol), np.ascontiguousarray(means)
        log_det, log_w = np.ascontiguousarray(log_det), np.ascontiguousarray(log_w)
        total = c_double()
        _native.check(self._lib.zk_gmm_estep(self._h, _p(prec_chol), _p(means), _p(log_det), _p(log_w), len(means),
                                             int(want_resp), byref(total)), "zk_gmm_estep")
        return total.value

    def resp_from_labels(self, k):
        _native.check(self._lib.zk_gmm_resp_from_labels(self._h, int(k)), "zk_gmm_resp_from_labels")

    def moments(self, component, shift, count=None):
        """Weighted moment matrix of one component, or (``count`` given) of ``count`` consecutive ones -> (count, D+1, D+1);
        up to eight components share one pass over the matrix (three when there are more than 47 features)."""
        shift = np.ascontiguousarray(shift)
        d1 = self.n_features + 1
        n = 1 if count is None else int(count)
        out = np.empty((n, d1, d1))
        # the matrix-core pass takes 2 .. 47 features (and ZK_WGRAM_VALU forces the vector kernel for A/B runs): otherwise three
        per_pass = 8 if 2 <= self.n_features <= 47 and not os.environ.get("ZK_WGRAM_VALU") else 3
        for c0 in range(0, n, per_pass):
            cc = min(per_pass, n - c0)
            _native.check(self._lib.zk_gmm_moments(self._h, int(component) + c0, cc, _p(shift), _p(out[c0:c0 + cc])), "zk_gmm_moments")
        return out[0] if count is None else out


    def gram(self, shift):
        """``sum_r [x_r - shift | 1]^T [x_r - shift | 1]`` (D+1, D+1), fixed summation order."""
        shift = np.ascontiguousarray(shift, dtype=np.float64)
        d1 = self.n_features + 1
        out = np.empty((d1, d1))
        _native.check(self._lib.zk_rows_gram(self._h, _p(shift), _p(out)), "zk_rows_gram")
        return out

    def project(self, mean, components):
        """``(X - mean) components^T`` -> (n_rows, k) on the host (page-locked); the matrix stays on the device."""
        lib, dev, n, d = self._lib, self.device, self.n_rows, self.n_features
        components = np.ascontiguousarray(components, dtype=np.float64)
        k = len(components)
        out = _native.pinned.empty((n, k))
        if n == 0:
            return out
        x_dev, bufs = lib.zk_rows_data(self._h), []

        def to_device(arr):
            arr = np.ascontiguousarray(arr, dtype=np.float64)
            ptr = c_void_p()
            _native.check(lib.zk_device_malloc(dev, arr.nbytes, byref(ptr)), "zk_device_malloc")
            bufs.append(ptr)
            _native.check(lib.zk_device_copy(dev, ptr, arr.ctypes.data_as(c_void_p), arr.nbytes, 1), "zk_device_copy")
            return ptr

        try:
            d_mean = to_device(mean)
            d_y = c_void_p()
            _native.check(lib.zk_device_malloc(dev, n * min(k, 16) * 8, byref(d_y)), "zk_device_malloc")
            bufs.append(d_y)
            for done in range(0, k, 16):                                     # at most 16 components per launch
                kk = min(16, k - done)
                d_comp = to_device(components[done:done + kk])
                _native.check(lib.zk_project_dev(dev, c_void_p(x_dev), n, d, d_mean, d_comp, kk, d_y, None), "zk_project_dev")
                part = out if kk == k else np.empty((n, kk))
                _native.check(lib.zk_device_synchronize(dev), "zk_device_synchronize")
                _native.check(lib.zk_device_copy(dev, part.ctypes.data_as(c_void_p), d_y, part.nbytes, 2), "zk_device_copy")
                if part is not out:
                    out[:, done:done + kk] = part
        finally:
            for ptr in bufs:
                lib.zk_device_free(dev, ptr)
        return out


class _NoRows:
    """A rank's block that holds no rows (a spread matrix with fewer blocks than ranks): every pass contributes nothing."""

    def __init__(self, n_features):
        self.n_rows, self.n_features = 0, int(n_features)

    def colsum(self):
        return np.zeros(self.n_features)

    def center_at(self, mean):
        return np.zeros(self.n_features), 0

    def fetch(self, idx, centred=True):
        return np.empty((0, self.n_features))

    def seed_step(self, cand, cand_sq, use_closest):
        return np.zeros(len(cand))

    def seed_pick(self, which, vals=()):
        return np.zeros(len(vals), dtype=np.int64)

    def reset_labels(self):
        pass

    def lloyd(self, centers, update=True):
        return np.zeros((len(centers), self.n_features)), np.zeros(len(centers)), 0

    def own_distance(self, centers):
        return np.empty(0)

    def labels(self):
        return np.empty(0, dtype=np.int32)

    def estep(self, prec_chol, means, log_det, log_w, want_resp=True):
        return 0.0

    def resp_from_labels(self, k):
        pass

    def moments(self, component, shift, count=None):
        d1 = self.n_features + 1
        return np.zeros((d1, d1) if count is None else (int(count), d1, d1))

    gram = lambda self, shift: np.zeros((self.n_features + 1, self.n_features + 1))

    def project(self, mean, components):
        return np.empty((0, len(components)))

    def close(self):
        pass


def _as_rows(X):
    if isinstance(X, (DeviceRows, _NoRows)):
        return X, False
    X = np.asarray(X)
    if X.ndim == 2 and X.shape[0] == 0 and X.shape[1] > 0:
        return _NoRows(X.shape[1]), True
    return DeviceRows(X), True


class _Shards:
    """The rows of one matrix spread over the ranks of a communicator (rank order = row order), or all of them here
    (``comm`` None).  Everything the clustering control flow needs from the other ranks is a sum of small arrays, taken in
    rank order on every rank (``comm.allgather_host``: the same bits everywhere), or a row fetched from its owner: no rank
    ever sees another rank's block of the matrix."""

    def __init__(self, rows, comm=None):
        self.rows, self.comm = rows, comm
        if comm is None or comm.world == 1:
            self.comm = None
            self.counts = np.array([rows.n_rows], dtype=np.int64)
            self.rank = 0
        else:
            self.counts = self.gather(np.array([rows.n_rows], dtype=np.int64))[:, 0]
            self.rank = comm.rank
        self.offsets = np.concatenate([[0], np.cumsum(self.counts)])
        self.total = int(self.offsets[-1])
        self._stats = None

    def gather(self, arr):
        """Every rank's copy of a small float64 / int64 array -> (world, ...)."""
        arr = np.ascontiguousarray(arr)
        if self.comm is None:
            return arr[None]
        return np.stack([np.frombuffer(b, dtype=arr.dtype).reshape(arr.shape) for b in self.comm.allgather_host(arr.tobytes())])

    def sum(self, arr):
        parts = self.gather(arr)
        out = parts[0].copy()
        for part in parts[1:]:
            out += part
        return out

    def center(self):
        """(mean, population variance, rows with a non-finite element) of the whole matrix; sets the centring shift."""
        cached = getattr(self.rows, "_center_cache", None) if self.comm is None else None
        if self._stats is None and cached is not None:
            self._stats = cached                                             # one block: the shift is already the block's mean
        if self._stats is None:
            mean = self.sum(self.rows.colsum()) / self.total
            sq, bad = self.rows.center_at(mean)
            self._stats = (mean, self.sum(sq) / self.total, int(self.sum(np.array([bad], dtype=np.int64))[0]))
            self.rows._center_cache = self._stats if self.comm is None else None
        return self._stats

    def check_finite(self):
        if self.center()[2]:
            raise ValueError("Input X contains NaN or infinity.")

    def fetch(self, global_ids):
        """Centred rows by global index, from whichever rank holds each."""
        ids = np.asarray(global_ids, dtype=np.int64).ravel()
        lo, hi = self.offsets[self.rank], self.offsets[self.rank + 1]
        mine = (ids >= lo) & (ids < hi)
        out = np.zeros((len(ids), self.rows.n_features))
        if mine.any():
            out[mine] = self.rows.fetch(ids[mine] - lo)
        return out if self.comm is None else self.sum(out)          # every row has exactly one owner: the others add zeros

    def search(self, which, vals, local_total):
        """``searchsorted(cumsum(closest), vals)`` over the whole matrix: the owner of each value follows from the ranks'
        totals, the position inside its block is found by that rank; also adopts candidate ``which`` as the closest-distance
        row on every rank."""
        vals = np.asarray(vals, dtype=np.float64)
        ends = np.cumsum(self.gather(np.array([local_total], dtype=np.float64))[:, 0])
        last_holder = np.flatnonzero(self.counts > 0)[-1]
        owner = np.minimum(np.searchsorted(ends, vals), last_holder)
        before = np.concatenate([[0.0], ends])[owner]
        mine = owner == self.rank
        found = np.full(len(vals), -1, dtype=np.int64)
        found[mine] = self.rows.seed_pick(which, (vals - before)[mine]) + self.offsets[self.rank]
        return found if self.comm is None else self.gather(found).max(axis=0)


# ---------------------------------------------------------------------------------------------------------- k-means
def _row_norms_sq(a):
    return np.einsum("ij,ij->i", a, a)                                       # sklearn.utils.extmath.row_norms(squared=True)


def _uniform_choice(random_state, n):
    """``random_state.choice(n, p=ones(n) / n)``: the same single draw from the generator and the same index (the cumulative
    sums NumPy would build are replayed by ``zk_uniform_choice_index`` without the three n-sized arrays)."""
    u = random_state.random_sample()
    idx = c_int64()
    _native.check(_native.load().zk_uniform_choice_index(int(n), float(u), byref(idx)), "zk_uniform_choice_index")
    return int(idx.value)


def _kmeans_plusplus(sh, n_clusters, random_state):
    """``_kmeans_plusplus`` (sklearn/cluster/_kmeans.py) on the centred matrix: same draws, distances on the device."""
    rows, n = sh.rows, sh.total
    n_local_trials = 2 + int(np.log(n_clusters))
    centers = np.empty((n_clusters, rows.n_features))
    indices = np.full(n_clusters, -1, dtype=int)
    center_id = _uniform_choice(random_state, n)                             # random_state.choice(n, p=weight / weight.sum())
    centers[0] = sh.fetch([center_id])[0]
    indices[0] = center_id
    local_pots = rows.seed_step(centers[:1], _row_norms_sq(centers[:1]), use_closest=False)
    current_pot = sh.sum(local_pots)[0]
    which = 0
    for c in range(1, n_clusters):
        rand_vals = random_state.uniform(size=n_local_trials) * current_pot
        candidate_ids = sh.search(which, rand_vals, local_pots[which])       # searchsorted(stable_cumsum(closest), rand_vals)
        cand = sh.fetch(candidate_ids)
        local_pots = rows.seed_step(cand, _row_norms_sq(cand), use_closest=True)
        pots = sh.sum(local_pots)
        which = int(np.argmin(pots))
        current_pot = pots[which]
        centers[c] = cand[which]
        indices[c] = candidate_ids[which]
    return centers, indices


def _relocate_empty_clusters(sh, centers_old, centers_new, counts):
    """``_relocate_empty_clusters_dense`` (sklearn/cluster/_k_means_common.pyx): the points farthest from their centres
    become the centres of the empty clusters; labels are left as they are.  (Several ranks: each offers its own farthest
    points and the farthest of all are taken in descending order -- scikit-learn's order among them is that of
    ``numpy.argpartition``, which only a single block can reproduce.)"""
    empty = np.where(counts == 0)[0]
    if len(empty) == 0:
        return
    rows, n_empty = sh.rows, len(empty)
    distances = rows.own_distance(centers_old)
    if sh.comm is None:
        far = np.argpartition(distances, -n_empty)[:-n_empty - 1:-1]
        old_ids = rows.labels()[far]
        points = rows.fetch(far)
    else:
        take = min(n_empty, rows.n_rows)
        local = np.argsort(distances)[::-1][:take]
        offer = np.full((n_empty, 3), -np.inf)                               # (distance, global index, label)
        offer[:take, 0] = distances[local]
        offer[:take, 1] = local + sh.offsets[sh.rank]
        offer[:take, 2] = rows.labels()[local]
        offers = sh.gather(offer).reshape(-1, 3)
        best = offers[np.argsort(-offers[:, 0], kind="stable")[:n_empty]]
        old_ids = best[:, 2].astype(np.int64)
        points = sh.fetch(best[:, 1].astype(np.int64))
    for new_id, old_id, x in zip(empty, old_ids, points):
        centers_new[old_id] -= x
        centers_new[new_id] = x
        counts[new_id] = 1.0
        counts[old_id] -= 1.0


def kmeans_fit(X, n_clusters, random_state=0, max_iter=300, tol=1e-4, comm=None):
    """``KMeans(n_clusters, random_state=random_state)`` with scikit-learn's defaults (k-means++ seeding, one run, Lloyd):
    returns ``(labels int32 (N), cluster_centers (k, D), n_iter)``.  ``X``: array or ``DeviceRows``.

    With ``comm`` (``mtflearn_amd.distributed.RcclComm`` / ``TorchComm``) ``X`` is this rank's block of rows of a matrix
    spread over the ranks in rank order -- e.g. the moments each GPU computed -- and the clustering is that of the whole
    matrix: the ranks exchange ``k x D`` sums per iteration, never rows of the matrix (no all-gather of the moments); the
    returned labels are this rank's block.  Every rank must pass the same ``random_state``."""
    from sklearn.utils import check_random_state
    rows, own = _as_rows(X)
    try:
        sh = _Shards(rows, comm)
        if not isinstance(n_clusters, (int, np.integer)) or n_clusters < 1:
            raise ValueError(f"The 'n_clusters' parameter of KMeans must be an int in the range [1, inf). Got {n_clusters!r} instead.")
        if sh.total < n_clusters:
            raise ValueError(f"n_samples={sh.total} should be >= n_clusters={n_clusters}.")
        if n_clusters > 256:
            raise ValueError("at most 256 clusters on the device")
        sh.check_finite()
        mean, var, _ = sh.center()
        tol_abs = np.mean(var) * tol                                         # _tolerance(X, tol)
        rs = check_random_state(random_state)
        centers, _ = _kmeans_plusplus(sh, n_clusters, rs)
        rows.reset_labels()
        strict = False
        n_iter = 0
        for n_iter in range(1, max_iter + 1):                                # _kmeans_single_lloyd
            sums, counts, n_changed = rows.lloyd(centers, update=True)
            if sh.comm is not None:
                sums, counts = sh.sum(sums), sh.sum(counts)
                n_changed = int(sh.sum(np.array([n_changed], dtype=np.int64))[0])
            present = np.array(counts, dtype=np.int64)                       # = bincount of this pass's labels, over all ranks
            _relocate_empty_clusters(sh, centers, sums, counts)
            centers_new = sums
            filled = counts > 0
            centers_new[filled] *= (1.0 / counts[filled])[:, None]           # _average_centers
            shift = np.sqrt(((centers_new - centers) ** 2).sum(axis=1))      # _center_shift
            centers = centers_new
            if n_changed == 0:
                strict = True
                break
            if (shift ** 2).sum() <= tol_abs:
                break
        if not strict:
            # labels consistent with the final centres; the pass's own counts say which clusters are populated (a host
            # bincount of the labels costs 3 ms at 4 M rows -- more than the rest of a two-iteration fit)
            _, counts, _ = rows.lloyd(centers, update=True)
            present = np.array(counts if sh.comm is None else sh.sum(counts), dtype=np.int64)
        labels = rows.labels()
        if np.count_nonzero(present) < n_clusters:
            from sklearn.exceptions import ConvergenceWarning
            warnings.warn(f"Number of distinct clusters ({np.count_nonzero(present)}) found smaller than n_clusters "
                          f"({n_clusters}). Possibly due to duplicate points in X.", ConvergenceWarning, stacklevel=2)
        return labels, centers + mean, n_iter
    finally:
        if own:
            rows.close()


def gather_labels(labels, comm):
    """Every rank's block of labels -> the labels of the whole matrix on every rank (4 bytes per row: at n_max 8 that is
    1 / 90 of what gathering the moments would move)."""
    if comm is None or comm.world == 1:
        return labels
    counts = [np.frombuffer(b, dtype=np.int64)[0] for b in comm.allgather_host(np.int64(len(labels)).tobytes())]
    padded = np.zeros(max(counts), dtype=labels.dtype)
    padded[:len(labels)] = labels
    return np.concatenate([np.frombuffer(b, dtype=labels.dtype)[:n] for b, n in zip(comm.allgather_host(padded.tobytes()), counts)])


def _relabel_by_size(lbs):
    """The reference's relabelling (``_clustering_functions.py:17-21`` / ``28-32``), with its dictionary built the same way:
    ``dict(zip(argsort(counts)[::-1], unique))`` applied to every label."""
    present = np.bincount(lbs)
    unique = np.flatnonzero(present).astype(lbs.dtype)                       # np.unique(lbs, return_counts=True)
    counts = present[unique]
    lbs_order = np.argsort(counts)[::-1]
    order_dict = dict(zip(lbs_order, unique))
    if all(u in order_dict for u in unique):
        lut = np.zeros(len(present), dtype=lbs.dtype)                        # the values are elements of `unique`
        for key, value in order_dict.items():
            lut[key] = value
        return lut[lbs]
    return np.vectorize(order_dict.get)(lbs)             # a label without a cluster: the reference's own call (raises)


def kmeans_lbs(X, n=None, random_state=0, comm=None):
    """Drop-in for ``mtflearn.clustering.kmeans_lbs`` (reference ``_clustering_functions.py:8-22``).  With ``comm``: ``X`` is
    this rank's block of rows (see ``kmeans_fit``); the labels of the WHOLE matrix are returned on every rank."""
    labels, _, _ = kmeans_fit(X, n, random_state=random_state, comm=comm)
    return _relabel_by_size(gather_labels(labels, comm))


def sort_lbs(lbs):
    """``mtflearn.clustering.sort_lbs`` (reference ``_clustering_functions.py:36-43``): labels renumbered by decreasing size."""
    unique_lbs, counts = np.unique(lbs, return_counts=True)
    unique_lbs = unique_lbs[np.argsort(counts)[::-1]]
    lut = dict(zip(unique_lbs.tolist(), range(len(unique_lbs))))
    return np.vectorize(lut.get)(lbs)


def normalize_xy(xy, low=0, high=1):
    """``mtflearn.clustering.normalize_xy`` (reference ``_clustering_functions.py:46-50``)."""
    vmax = xy.max()
    vmin = xy.min()
    return (xy - vmin) / (vmax - vmin) * (high - low) + low


def seg_lbs(xy, size=256, t=0.01):
    """``mtflearn.clustering.seg_lbs`` (reference ``_clustering_functions.py:52-64``): the points of a 2-D layout (e.g.
    ``ForceGraph8``'s) rasterised into a ``size`` x ``size`` image, dilated by a radius-3 disk, segmented into connected
    components, labels renumbered by decreasing size.  A few thousand pixel operations: host code.  The reference calls
    scikit-image's ``dilation(aa, disk(3))`` and ``label(aa)`` (absent from this image); the same operations are taken from
    SciPy here -- grey dilation with the ``x^2 + y^2 <= 9`` footprint, 8-connected components numbered in raster order.
    Parity unpinned (nothing of scikit-image can run here)."""
    from scipy import ndimage as ndi
    xy_norm = normalize_xy(xy, -size * 0.9 // 2, size * 0.9 // 2) + size // 2
    xy_ = np.round(xy_norm).astype(int)
    x, y = xy_.T
    aa = np.zeros((size, size))
    s = 3
    aa[y, x] = 1
    grid = np.arange(-s, s + 1)
    disk = (grid[:, None] ** 2 + grid[None, :] ** 2 <= s * s)
    aa = ndi.grey_dilation(aa, footprint=disk, mode="constant", cval=0.0)
    lbs_img, _ = ndi.label(aa, structure=np.ones((3, 3)))
    lbs = lbs_img[y, x]
    lbs = lbs - lbs.min()
    return sort_lbs(lbs)


# ------------------------------------------------------------------------------------------------- Gaussian mixture
_CHOL_ERROR = ("Fitting the mixture model failed because some components have ill-defined empirical covariance (for instance "
               "caused by singleton or collapsed samples). Try to decrease the number of components, increase reg_covar, or "
               "scale the input data.")


def _gaussian_parameters(sh, k, shift, reg_covar, covariance_type):
    """``_estimate_gaussian_parameters`` (sklearn/mixture/_gaussian_mixture.py) from the device's weighted moments about
    ``shift``: (nk, means, covariances)."""
    rows = sh.rows
    d = rows.n_features
    eps10 = 10 * np.finfo(np.float64).eps
    nk, means = np.empty(k), np.empty((k, d))
    scatter = np.empty((k, d, d))                                             # sum_r w (x - mean)(x - mean)^T
    grams = sh.sum(rows.moments(0, shift, count=k))                          # three components per pass over the matrix
    for c in range(k):
        g = grams[c]
        a, b, n0 = g[:d, :d], g[d, :d], g[d, d]
        nk[c] = n0 + eps10
        mt = b / nk[c]
        means[c] = shift + mt
        scatter[c] = a - np.outer(mt, b) - np.outer(b, mt) + n0 * np.outer(mt, mt)
    if covariance_type == "full":
        cov = scatter / nk[:, None, None]
        cov[:, np.arange(d), np.arange(d)] += reg_covar
    elif covariance_type == "tied":
        cov = scatter.sum(axis=0) / nk.sum()
        cov[np.arange(d), np.arange(d)] += reg_covar
    elif covariance_type == "diag":
        cov = scatter[:, np.arange(d), np.arange(d)] / nk[:, None] + reg_covar
    elif covariance_type == "spherical":
        cov = (scatter[:, np.arange(d), np.arange(d)] / nk[:, None] + reg_covar).mean(axis=1)
    else:
        raise ValueError(f"The 'covariance_type' parameter of GaussianMixture must be a str among "
                         f"{{'full', 'tied', 'diag', 'spherical'}}. Got {covariance_type!r} instead.")
    return nk, means, cov


_BLAS_CONTROLLER = None


def _one_blas_thread():
    """Context in which the host's BLAS / LAPACK runs on one thread.  The matrices between the passes are D x D: on the 256-CPU
    host of a GPU box a threaded ``scipy.linalg.solve_triangular`` of a 45 x 45 system takes 2.2 ms (waking its workers),
    30 of them were 66 of the 72 ms of a four-iteration ``gmm_fit_predict`` on 4 M rows; single-threaded it takes ~20 us.
    (Same LAPACK routines: every right-hand-side column is solved in the same order either way.)"""
    global _BLAS_CONTROLLER
    try:
        if _BLAS_CONTROLLER is None:
            from threadpoolctl import ThreadpoolController
            _BLAS_CONTROLLER = ThreadpoolController()
        return _BLAS_CONTROLLER.limit(limits=1, user_api="blas")
    except Exception:                                                        # no threadpoolctl: the plain, threaded calls
        import contextlib
        return contextlib.nullcontext()


def _precision_cholesky(cov, covariance_type, k, d):
    """``_compute_precision_cholesky`` -> upper-triangular factors (k, D, D) for the E-step kernel + ``log_det``."""
    with _one_blas_thread():
        return _precision_cholesky_host(cov, covariance_type, k, d)


def _precision_cholesky_host(cov, covariance_type, k, d):
    from scipy import linalg
    out = np.zeros((k, d, d))

    def chol(c):
        try:
            low = linalg.cholesky(c, lower=True)
        except linalg.LinAlgError:
            raise ValueError(_CHOL_ERROR)
        return linalg.solve_triangular(low, np.eye(d), lower=True).T

    if covariance_type == "full":
        for c in range(k):
            out[c] = chol(cov[c])
    elif covariance_type == "tied":
        out[:] = chol(cov)
    else:
        if np.any(np.less_equal(cov, 0.0)):
            raise ValueError(_CHOL_ERROR)
        diag = 1.0 / np.sqrt(cov)
        out[:, np.arange(d), np.arange(d)] = diag if covariance_type == "diag" else diag[:, None]
    log_det = np.sum(np.log(out[:, np.arange(d), np.arange(d)]), axis=1)     # _compute_log_det_cholesky
    return out, log_det


def gmm_fit_predict(X, n_components, covariance_type="full", random_state=0, tol=1e-3, reg_covar=1e-6, max_iter=100, comm=None):
    """``GaussianMixture(n_components, covariance_type=..., random_state=...).fit(X).predict(X)`` with scikit-learn's defaults
    (k-means initialisation, one run): returns ``(labels int32 (N), n_iter, converged)``.  With ``comm``: ``X`` is this rank's
    block of rows and the mixture is that of the whole matrix (see ``kmeans_fit``): the ranks exchange the lower bound and
    ``k`` weighted ``(D + 1) x (D + 1)`` moment matrices per EM iteration."""
    from sklearn.utils import check_random_state
    rows, own = _as_rows(X)
    try:
        sh = _Shards(rows, comm)
        k, d, n = int(n_components), rows.n_features, sh.total
        if n < 2:
            raise ValueError(f"Found array with {n} sample(s) (shape=({n}, {d})) while a minimum of 2 is required by GaussianMixture.")
        if n < k:
            raise ValueError(f"Expected n_samples >= n_components but got n_components = {k}, n_samples = {n}")
        if k > 64:
            raise ValueError("at most 64 mixture components on the device")
        sh.check_finite()
        shift = sh.center()[0]
        rs = check_random_state(random_state)
        kmeans_fit(rows, k, random_state=rs, comm=comm)                      # _initialize_parameters, init_params='kmeans'
        rows.resp_from_labels(k)
        weights, means, cov = _gaussian_parameters(sh, k, shift, reg_covar, covariance_type)
        weights /= n
        prec, log_det = _precision_cholesky(cov, covariance_type, k, d)
        lower_bound, converged, n_iter = -np.inf, False, 0
        for n_iter in range(1, max_iter + 1):
            prev = lower_bound
            lower_bound = sh.sum(np.array([rows.estep(prec, means, log_det, np.log(weights))]))[0] / n
            weights, means, cov = _gaussian_parameters(sh, k, shift, reg_covar, covariance_type)
            weights /= weights.sum()
            prec, log_det = _precision_cholesky(cov, covariance_type, k, d)
            if abs(lower_bound - prev) < tol:
                converged = True
                break
        if not converged and max_iter > 0:
            from sklearn.exceptions import ConvergenceWarning
            warnings.warn("Best performing initialization did not converge. Try different init parameters, or increase "
                          "max_iter, tol, or check for degenerate data.", ConvergenceWarning, stacklevel=2)
        rows.estep(prec, means, log_det, np.log(weights), want_resp=False)   # the final E step = predict(X)
        return rows.labels(), n_iter, converged
    finally:
        if own:
            rows.close()


def gmm_lbs(X, n, type="full", ramdom_state=0, comm=None):
    """Drop-in for ``mtflearn.clustering.gmm_lbs`` (reference ``_clustering_functions.py:25-33``; the keyword is spelt
    ``ramdom_state`` there).  With ``comm``: ``X`` is this rank's block, the labels of the whole matrix are returned."""
    labels, _, _ = gmm_fit_predict(X, n, covariance_type=type, random_state=ramdom_state, comm=comm)
    return _relabel_by_size(gather_labels(labels, comm).astype(np.intp))     # predict() = argmax: numpy's index type
