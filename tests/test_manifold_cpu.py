"""The host side of the manifold consumer (no GPU): ``zk_force_layout_stage`` -- compiled host code in ``libzernike_hip.so``,
like the reference's numba-compiled ``optimize_stage`` -- against the oracle's statement-by-statement Python restatement of
``mtflearn/manifold/force_relaxed.py:174-282``.  Same operations in the same order: the layouts must be BIT-identical, the
random states too.  (Parity with the reference itself is unpinned: numba is not installed, see oracle/manifold_oracle.py.)"""
import numpy as np
import pytest
from scipy import sparse


@pytest.fixture(scope="module")
def mo():
    from oracle import manifold_oracle
    return manifold_oracle


def _random_problem(rng, n, k):
    nbrs = np.stack([np.concatenate([[i], rng.choice(np.delete(np.arange(n), i), k - 1, replace=False)]) for i in range(n)])
    P = rng.random((n, k))
    P[:, 0] = 0.0
    graph = sparse.csr_matrix((P.ravel(), nbrs.ravel(), range(0, n * k + 1, k)), shape=(n, n))
    graph = graph + graph.T - graph.multiply(graph.T)
    xy = rng.uniform(-10, 10, (n, 2))
    return nbrs.astype(np.int64), graph, xy


def test_tau_rand_int_sequence(mo):
    from mtflearn_amd import manifold as M
    rng = np.random.default_rng(0)
    for _ in range(20):
        seed = rng.integers(mo.INT32_MIN, mo.INT32_MAX, 3).astype(np.int64)
        # drive the native generator through a stage with no pairs' attraction effect: one node pair, weight 0, and read the
        # state afterwards; the oracle advances its copy by the same number of draws
        state = seed.copy()
        xy = np.zeros((3, 2))
        pairs = np.zeros(1, dtype=[('node1', int), ('node2', int), ('weight', np.float64)])
        M.optimize_layout(4, xy, pairs, 7, np.array([[0], [1], [2]], dtype=np.int64), 1.0, (0, 2, 1, 1), (2, 4, 5, 2), state, 0.5)
        ref = [int(v) for v in seed]
        for _ in range(4 * 7):
            mo.tau_rand_int(ref)
        assert [int(v) for v in state] == ref


@pytest.mark.parametrize("n,k,iters,neg,params", [
    (40, 4, 12, 3, ((0, 2, 1, 1), (2, 4, 5, 2))),
    (90, 6, 20, 10, ((0, 2, 1, 1), (2, 4, 5, 2))),
    (25, 3, 9, 2, ((1.5, 3, 0.7, 2.5), (2, 2.5, 5, 0.3))),
])
def test_layout_stage_is_bit_identical_to_the_restatement(mo, n, k, iters, neg, params):
    from mtflearn_amd import manifold as M
    rng = np.random.default_rng(n)
    nbrs, graph, xy0 = _random_problem(rng, n, k)
    seed = np.random.RandomState(5).randint(mo.INT32_MIN, mo.INT32_MAX, 3).astype(np.int64)
    for divide in (0.5, 0.3):
        xy_ref, state_ref = xy0.copy(), [int(v) for v in seed]
        logs_ref = mo.optimize_layout(iters, xy_ref, mo.compute_pairs(graph), neg, nbrs, 1.0, np.array(params[0]), np.array(params[1]),
                                      state_ref, divide)
        xy, state = xy0.copy(), seed.copy()
        pairs = M.compute_pairs(graph)
        np.testing.assert_array_equal(pairs['node1'], mo.compute_pairs(graph)[0])
        logs = M.optimize_layout(iters, xy, pairs, neg, nbrs, 1.0, np.array(params[0]), np.array(params[1]), state, divide)
        np.testing.assert_array_equal(xy, xy_ref)
        assert [int(v) for v in state] == state_ref
        assert len(logs) == len(logs_ref)
        for a, b in zip(logs, logs_ref):
            np.testing.assert_array_equal(a, b)
        assert np.isfinite(xy).all() and not np.array_equal(xy, xy0)


def test_layout_stage_rejects_bad_indices():
    from mtflearn_amd import manifold as M
    pairs = np.zeros(1, dtype=[('node1', int), ('node2', int), ('weight', np.float64)])
    pairs['node2'] = 9
    with pytest.raises(RuntimeError, match="out of range"):
        M.optimize_layout(2, np.zeros((3, 2)), pairs, 1, np.zeros((3, 1), dtype=np.int64), 1.0, (0, 2, 1, 1), (2, 4, 5, 2),
                          np.array([1, 2, 3], dtype=np.int64), 0.5)
