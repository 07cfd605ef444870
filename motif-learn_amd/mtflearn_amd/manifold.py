"""The manifold consumer of the moment matrix -- the mirror of ``mtflearn.manifold``.

``ForceGraph8`` keeps the constructor, attributes and ``fit`` / ``fit_transform`` of the reference class
(``mtflearn/manifold/force_relaxed.py:285-366``).  Where its stages run:

* ``compute_graph`` (``:67-86``): the neighbour search -- scikit-learn's brute-force ``NearestNeighbors(metric='correlation')``
  in the reference, the O(N^2 D) part -- and the per-row bisection of ``calculate_asymmetric_Pij`` (``:17-52``) run on the GPU
  over the resident matrix (``zk_rows_knn_correlation``); the sparse symmetrisation ``calculate_graph`` (``:55-64``) is the
  reference's own three SciPy calls.
* ``init_layout`` (``:89-116``): the PCA initialisation runs on the GPU (``mtflearn_amd.features.pca`` on the same resident
  matrix); the random one draws from the instance's ``RandomState`` as the reference does.
* ``optimize_layout`` (``:269-282``): ONE strictly sequential loop in the reference -- every pair update moves two nodes the
  next pair reads, the repulsion partners come from one running ``tau_rand_int`` state -- compiled by numba there, compiled host
  code here (``zk_force_layout_stage`` in ``libzernike_hip.so``), the same operations in the same order.  There is nothing for a
  GPU to do in it without changing the algorithm.

Only ``metric='correlation'`` (the reference's default and the one its notebooks use) has a device neighbour search; other
metrics raise.  The reference prints progress lines from ``init_layout`` / ``calculate_graph``; they are kept behind ``verbose``.
"""
from __future__ import annotations

from ctypes import POINTER, c_double, c_int64

import numpy as np
from scipy import sparse
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_random_state

from . import _native
from .clustering import DeviceRows

__all__ = ["ForceGraph8", "compute_graph", "calculate_graph", "init_layout", "compute_pairs", "optimize_layout"]

MACHINE_EPSILON = np.finfo(np.double).eps
INT32_MIN = np.iinfo(np.int32).min + 1
INT32_MAX = np.iinfo(np.int32).max - 1


def calculate_graph(Pij, ind, set_op_mix_ratio=1.0, verbose=0):
    """Symmetric weights as a sparse matrix (reference ``force_relaxed.py:55-64``, the same SciPy calls)."""
    if verbose:
        print('Construct graph from data...')
    n_samples, k = Pij.shape
    P = sparse.csr_matrix((Pij.ravel(), ind.ravel(), range(0, n_samples * k + 1, k)), shape=(n_samples, n_samples))
    prod = P.multiply(P.T)
    return set_op_mix_ratio * (P + P.T - prod) + (1 - set_op_mix_ratio) * prod


def _knn_affinities(rows, n_neighbors, local_connectivity, perplexity):
    n = rows.n_rows
    ind = np.empty((n, n_neighbors), dtype=np.int64)
    dist, P = np.empty((n, n_neighbors)), np.empty((n, n_neighbors))
    _native.check(rows._lib.zk_rows_knn_correlation(rows._h, int(n_neighbors), int(local_connectivity), float(perplexity),
                                                    ind.ctypes.data_as(POINTER(c_int64)), dist.ctypes.data_as(POINTER(c_double)),
                                                    P.ctypes.data_as(POINTER(c_double))), "zk_rows_knn_correlation")
    return dist, ind, P


def compute_graph(X, n_neighbors, metric, perplexity=None, local_connectivity=1, set_op_mix_ratio=1.0):
    """(sparse symmetric graph, neighbour indices) of the rows of ``X`` (array or ``DeviceRows``); reference ``:67-86``."""
    if metric != "correlation":
        raise ValueError(f"only metric='correlation' has a device neighbour search (got {metric!r})")
    if perplexity is None:
        perplexity = n_neighbors
    rows, own = (X, False) if isinstance(X, DeviceRows) else (DeviceRows(X), True)
    try:
        if n_neighbors > rows.n_rows:
            raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors}, n_samples_fit = {rows.n_rows}, "
                             f"n_samples = {rows.n_rows}")
        _, ind, P_ij = _knn_affinities(rows, n_neighbors, local_connectivity, perplexity)
    finally:
        if own:
            rows.close()
    return calculate_graph(Pij=P_ij, ind=ind, set_op_mix_ratio=set_op_mix_ratio, verbose=0), ind


def init_layout(X, random_state, dim=2, init_mode='pca', verbose=False):
    """Initial layout (reference ``:89-116``): PCA scores scaled to +-10, or uniform random in [-10, 10)."""
    n = X.n_rows if isinstance(X, DeviceRows) else np.shape(X)[0]
    if init_mode == 'random':
        if verbose:
            print('Initialize {}-d embedding using random layout...'.format(dim))
        return check_random_state(random_state).uniform(low=-10.0, high=10.0, size=(n, dim))
    if init_mode == 'pca':
        if verbose:
            print('Initialize {}-d embedding using PCA layout...'.format(dim))
        from .features.consumers import pca
        X_pca = np.array(pca(X, n_components=dim))
        return X_pca / np.abs(X_pca).max() * 10


def compute_pairs(graph):
    """Structured array of the graph's non-zero entries in COO order (reference ``:150-171``)."""
    pair_dtype = np.dtype([('node1', int), ('node2', int), ('weight', np.float64)])
    g = sparse.coo_matrix(graph)
    pairs = np.empty(g.nnz, dtype=pair_dtype)
    pairs['node1'], pairs['node2'], pairs['weight'] = g.row, g.col, g.data
    return pairs


def _stage(num_iterations, xy, pairs, force_params, num_negative_samples, nbrs_ind, learning_rate, rng_states, logs):
    lib = _native.load()
    logs.append(xy.copy())
    if num_iterations <= 0:
        return logs
    node1 = np.ascontiguousarray(pairs['node1'], dtype=np.int64)
    node2 = np.ascontiguousarray(pairs['node2'], dtype=np.int64)
    weight = np.ascontiguousarray(pairs['weight'], dtype=np.float64)
    nbrs = np.ascontiguousarray(nbrs_ind, dtype=np.int64)
    fp = np.ascontiguousarray(force_params, dtype=np.float64)
    log = np.empty((num_iterations,) + xy.shape)
    I64, F64 = POINTER(c_int64), POINTER(c_double)
    _native.check(lib.zk_force_layout_stage(xy.ctypes.data_as(F64), xy.shape[0], node1.ctypes.data_as(I64), node2.ctypes.data_as(I64),
                                            weight.ctypes.data_as(F64), len(weight), nbrs.ctypes.data_as(I64), nbrs.shape[1],
                                            int(num_iterations), fp.ctypes.data_as(F64), int(num_negative_samples),
                                            float(learning_rate), rng_states.ctypes.data_as(I64), log.ctypes.data_as(F64)),
                  "zk_force_layout_stage")
    logs.extend(log)
    return logs


def optimize_layout(num_iterations, xy, pairs, num_negative_samples, nbrs_ind, learning_rate, force_params1, force_params2,
                    rng_states, divide):
    """Two stages of the force-directed sweep (reference ``:269-282``); ``xy`` (n, 2) float64 C-contiguous and ``rng_states``
    (3 int64) are updated in place; returns the reference's ``logs`` list (the layout before, and after every sweep)."""
    logs = [xy.copy()]
    logs = _stage(int(num_iterations * divide), xy, pairs, force_params1, num_negative_samples, nbrs_ind, learning_rate, rng_states, logs)
    logs = _stage(int(num_iterations * (1 - divide)), xy, pairs, force_params2, num_negative_samples, nbrs_ind, learning_rate,
                  rng_states, logs)
    return logs


class ForceGraph8(TransformerMixin, BaseEstimator):
    """Drop-in for ``mtflearn.manifold.ForceGraph8`` (reference ``force_relaxed.py:285-366``): same parameters, attributes
    (``graph``, ``nbrs_ind``, ``pts``, ``pairs``, ``rng_states``, ``logs``, ``y``) and methods."""

    def __init__(self, X=None, n_neighbors=10, metric='correlation', local_connectivity=1, random_state=48, init_mode='pca',
                 num_negative_samples=10, edge_weight_influence=1.0, learning_rate=1.0, num_iterations=100,
                 force_params1=(0, 2, 1, 1), force_params2=(2, 4, 5, 2), divide=0.5, verbose=False):
        self.X = X
        self.n_neighbors = n_neighbors
        self.metric = metric
        self.local_connectivity = local_connectivity
        self.random_state = check_random_state(random_state)
        self.init_mode = 'random' if init_mode is None else init_mode
        self.edge_weight_influence = edge_weight_influence
        self.num_negative_samples = num_negative_samples
        self.force_params1 = np.array(force_params1)
        self.force_params2 = np.array(force_params2)
        self.learning_rate = learning_rate
        self.num_iterations = num_iterations
        self.divide = divide
        self.verbose = verbose
        self.graph = None
        self.pts = None
        self.nodes = None
        self.pairs = None
        self.y = None
        self.logs = []

    def fit(self, X, y=None):
        rows, own = (X, False) if isinstance(X, DeviceRows) else (DeviceRows(X), True)     # one upload serves kNN and PCA
        try:
            self.graph, self.nbrs_ind = compute_graph(rows, self.n_neighbors, self.metric, None, self.local_connectivity, 1.0)
            self.pts = init_layout(rows, random_state=self.random_state, dim=2, init_mode=self.init_mode, verbose=self.verbose)
        finally:
            if own:
                rows.close()
        node_dtype = np.dtype([('x', np.float64), ('y', np.float64)])
        xy = np.ascontiguousarray(self.pts, dtype=np.float64).copy()
        self.pairs = compute_pairs(self.graph)
        self.rng_states = self.random_state.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
        self.logs = optimize_layout(self.num_iterations, xy, self.pairs, self.num_negative_samples, self.nbrs_ind,
                                    self.learning_rate, self.force_params1, self.force_params2, self.rng_states, self.divide)
        self.nodes = xy.view(node_dtype).reshape(-1)                 # the reference's structured array of (x, y) records
        self.y = xy.copy()

    def fit_transform(self, X, y=None):
        self.fit(X)
        return self.y
