"""CPU restatement of the reference's ForceGraph8 pipeline (``mtflearn/manifold/force_relaxed.py``) -- TEST INFRASTRUCTURE.

Only ``tests/`` may import this module; the product (``mtflearn_amd.manifold``) never does.

**Parity unpinned.**  The reference module imports numba (``@numba.njit(fastmath=True)`` on its optimiser), which is not
installed in this image, so the reference itself cannot be run here and no golden vectors exist; the reference holds no tests
for this path either.  What follows restates its algorithm statement by statement in plain Python / NumPy with IEEE arithmetic
in source order (``fastmath`` lets numba reassociate; its exact roundings are not knowable without running it).  Each function
cites the lines it follows.  Pure-Python loops: small cases only.
"""
import math

import numpy as np
from scipy import sparse

MACHINE_EPSILON = np.finfo(np.double).eps                       # force_relaxed.py:11
INT32_MIN = np.iinfo(np.int32).min + 1                          # :13
INT32_MAX = np.iinfo(np.int32).max - 1                          # :14


def calculate_asymmetric_Pij(dist_nn, perplexity=30, local_conectivity=1):
    """``force_relaxed.py:17-52``."""
    rho = dist_nn[:, local_conectivity][:, np.newaxis]
    d_ = dist_nn - rho
    d_[d_ < 0] = 0
    tolerance = 1e-5
    target = np.log2(perplexity)
    n_steps = 100
    beta_list = []
    for row in d_:
        beta_min, beta_max, beta = 0.0, np.inf, 1.0
        for n in np.arange(n_steps):
            sum_Pi = np.exp(-row[1:] * beta).sum()
            if np.abs(sum_Pi - target) < tolerance:
                beta_list.append(beta)
                break
            if sum_Pi - target > 0:
                beta_min = beta
                beta = beta * 2.0 if beta_max == np.inf else (beta + beta_max) / 2.0
            else:
                beta_max = beta
                beta = (beta + beta_min) / 2.0
            if n == (n_steps - 1):
                beta_list.append(beta)
    beta_list = np.array(beta_list)[:, np.newaxis]
    P_ij = np.exp(-d_ * beta_list)
    P_ij[P_ij < MACHINE_EPSILON] = MACHINE_EPSILON
    P_ij[:, 0] = 0.0
    return P_ij


def calculate_graph(Pij, ind, set_op_mix_ratio=1.0):
    """``force_relaxed.py:55-64``."""
    n_samples, k = Pij.shape
    P = sparse.csr_matrix((Pij.ravel(), ind.ravel(), range(0, n_samples * k + 1, k)), shape=(n_samples, n_samples))
    prod = P.multiply(P.T)
    return set_op_mix_ratio * (P + P.T - prod) + (1 - set_op_mix_ratio) * prod


def compute_graph(X, n_neighbors, metric, perplexity=None, local_connectivity=1, set_op_mix_ratio=1.0):
    """``force_relaxed.py:67-86`` (scikit-learn's brute-force neighbour search, as there)."""
    from sklearn.neighbors import NearestNeighbors
    if perplexity is None:
        perplexity = n_neighbors
    knn = NearestNeighbors(algorithm="auto", n_neighbors=n_neighbors, metric=metric).fit(X)
    d, ind = knn.kneighbors(X, n_neighbors=n_neighbors)
    P_ij = calculate_asymmetric_Pij(dist_nn=d, perplexity=perplexity, local_conectivity=local_connectivity)
    return calculate_graph(Pij=P_ij, ind=ind, set_op_mix_ratio=set_op_mix_ratio), ind


def init_layout(X, random_state, dim=2, init_mode="pca"):
    """``force_relaxed.py:89-116``."""
    from sklearn.decomposition import PCA
    from sklearn.utils import check_random_state
    if init_mode == "random":
        return check_random_state(random_state).uniform(low=-10.0, high=10.0, size=(X.shape[0], dim))
    X_pca = PCA(n_components=dim).fit_transform(X)
    return X_pca / np.abs(X_pca).max() * 10


def compute_pairs(graph):
    """``force_relaxed.py:150-171``: (node1, node2, weight) of the non-zero entries in COO order."""
    g = graph.tocoo()
    return np.asarray(g.row, dtype=np.int64), np.asarray(g.col, dtype=np.int64), np.asarray(g.data, dtype=np.float64)


def clip(val):
    """``force_relaxed.py:201-208``."""
    return 4.0 if val > 4.0 else (-4.0 if val < -4.0 else val)


def tau_rand_int(state):
    """``force_relaxed.py:210-233`` on a list of three Python ints holding the int64 state; returns the int32 value."""
    state[0] = (((state[0] & 4294967294) << 12) & 0xFFFFFFFF) ^ ((((state[0] << 13) & 0xFFFFFFFF) ^ state[0]) >> 19)
    state[1] = (((state[1] & 4294967288) << 4) & 0xFFFFFFFF) ^ ((((state[1] << 2) & 0xFFFFFFFF) ^ state[1]) >> 25)
    state[2] = (((state[2] & 4294967280) << 17) & 0xFFFFFFFF) ^ ((((state[2] << 3) & 0xFFFFFFFF) ^ state[2]) >> 11)
    v = (state[0] ^ state[1] ^ state[2]) & 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v                 # the "i4" return type


def optimize_stage(num_iterations, xy, pairs, force_params, num_negative_samples, nbrs_ind, learning_rate, rng_states, logs):
    """``force_relaxed.py:236-266``; ``xy`` (n, 2) float64 and ``rng_states`` (list of 3 ints) are updated in place."""
    num_nodes = len(xy)
    lr = learning_rate
    N, M, alpha, beta = (float(v) for v in force_params)
    logs.append(xy.copy())
    node1s, node2s, weights = pairs
    for n in range(0, num_iterations):
        for a, b, weight in zip(node1s.tolist(), node2s.tolist(), weights.tolist()):
            x_dist = xy[a, 0] - xy[b, 0]                             # apply_attraction_force, :174-184
            y_dist = xy[a, 1] - xy[b, 1]
            distance = float(np.hypot(x_dist, y_dist))              # np.hypot = libm's hypot (math.hypot is CPython's own)
            force = alpha / (math.pow(distance, N) + 1)
            fx = clip(x_dist * force) * lr * weight
            fy = clip(y_dist * force) * lr * weight
            xy[a, 0] -= fx
            xy[a, 1] -= fy
            xy[b, 0] += fx
            xy[b, 1] += fy
            for _ in range(num_negative_samples):
                rand_ind = tau_rand_int(rng_states) % num_nodes
                if all(ind != rand_ind for ind in nbrs_ind[a]):
                    x_dist = xy[a, 0] - xy[rand_ind, 0]              # apply_repulsion_force, :187-197
                    y_dist = xy[a, 1] - xy[rand_ind, 1]
                    distance = float(np.hypot(x_dist, y_dist))
                    force = beta / (math.pow(distance, M) + 1)
                    fx = clip(x_dist * force) * lr
                    fy = clip(y_dist * force) * lr
                    xy[a, 0] += fx
                    xy[a, 1] += fy
                    xy[rand_ind, 0] -= fx
                    xy[rand_ind, 1] -= fy
        lr = learning_rate * (1.0 - (float(n) / float(num_iterations)))
        logs.append(xy.copy())
    return logs


def optimize_layout(num_iterations, xy, pairs, num_negative_samples, nbrs_ind, learning_rate, force_params1, force_params2,
                    rng_states, divide):
    """``force_relaxed.py:269-282``."""
    logs = [xy.copy()]
    logs = optimize_stage(int(num_iterations * divide), xy, pairs, force_params1, num_negative_samples, nbrs_ind, learning_rate,
                          rng_states, logs)
    logs = optimize_stage(int(num_iterations * (1 - divide)), xy, pairs, force_params2, num_negative_samples, nbrs_ind,
                          learning_rate, rng_states, logs)
    return logs


def force_graph8(X, n_neighbors=10, metric="correlation", local_connectivity=1, random_state=48, init_mode="pca",
                 num_negative_samples=10, learning_rate=1.0, num_iterations=100, force_params1=(0, 2, 1, 1),
                 force_params2=(2, 4, 5, 2), divide=0.5):
    """``ForceGraph8.__init__`` + ``fit`` (``force_relaxed.py:285-362``): returns (y, logs, graph, nbrs_ind)."""
    from sklearn.utils import check_random_state
    rs = check_random_state(random_state)
    graph, nbrs_ind = compute_graph(X, n_neighbors, metric, None, local_connectivity, 1.0)
    xy = np.array(init_layout(X, random_state=rs, dim=2, init_mode=init_mode or "random"), dtype=np.float64)
    pairs = compute_pairs(graph)
    rng_states = [int(v) for v in rs.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)]
    logs = optimize_layout(num_iterations, xy, pairs, num_negative_samples, nbrs_ind, learning_rate, np.array(force_params1),
                           np.array(force_params2), rng_states, divide)
    return xy, logs, graph, nbrs_ind
