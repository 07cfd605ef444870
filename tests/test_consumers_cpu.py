"""Host pieces of the clustering consumers (no GPU): the reference's relabelling by cluster size against labels captured
from the reference (tests/golden/consumers_golden.npz, oracle/make_golden_consumers.py)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumers_golden.npz"))


def test_sort_lbs_matches_reference(golden):
    from mtflearn_amd.clustering import sort_lbs
    got = sort_lbs(golden["sort_in"])
    assert got.dtype == golden["sort_out"].dtype
    np.testing.assert_array_equal(got, golden["sort_out"])


def test_relabel_by_size_is_the_reference_mapping(golden):
    """Applied to scikit-learn's own labels (what the reference feeds it) the mapping reproduces the reference's output."""
    from sklearn.cluster import KMeans
    from mtflearn_amd.clustering import _relabel_by_size
    for key, n in (("Xa", 4), ("Xb", 5)):
        lbs = KMeans(n_clusters=n, random_state=0).fit(golden[key]).labels_
        np.testing.assert_array_equal(_relabel_by_size(lbs), golden[f"kmeans_{key}_{n}"])
    # the largest cluster becomes 0, sizes descend
    out = _relabel_by_size(np.array([2, 2, 2, 0, 1, 1], dtype=np.int32))
    np.testing.assert_array_equal(out, [0, 0, 0, 2, 1, 1])
    # a missing label makes the reference's dictionary lookup return None, on which its np.vectorize call raises:
    # reproduced (same call), not repaired
    with pytest.raises(TypeError):
        _relabel_by_size(np.array([0, 0, 2], dtype=np.int32))


def test_no_gpu_means_an_error_not_a_fallback(golden):
    from mtflearn_amd import _native
    from mtflearn_amd.clustering import kmeans_lbs
    if _native.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        kmeans_lbs(golden["Xa"], 4)


def test_seg_lbs_segments_a_layout():
    """``seg_lbs`` (host only): three well-separated point clouds of different sizes come back as labels 0, 1, 2 by size."""
    from mtflearn_amd.clustering import seg_lbs, normalize_xy
    rng = np.random.default_rng(0)
    sizes = (300, 150, 60)
    xy = np.concatenate([rng.standard_normal((n, 2)) * 0.4 + c for n, c in zip(sizes, ((0, 0), (8, 1), (3, 9)))])
    lbs = seg_lbs(xy)
    start = 0
    for want, n in enumerate(sizes):
        assert (lbs[start:start + n] == want).mean() > 0.9
        start += n
    np.testing.assert_allclose(normalize_xy(np.array([2.0, 4.0, 6.0]), -1, 1), [-1.0, 0.0, 1.0])


def test_uniform_choice_is_numpy_choice():
    """The replay of ``RandomState.choice(n, p=uniform)``: same index, same generator state afterwards, for sizes around
    powers of two, primes, and draws at the ends of the range."""
    from mtflearn_amd.clustering import _uniform_choice
    sizes = [1, 2, 3, 7, 10, 63, 64, 65, 1000, 4097, 65537, 100003, 1 << 20, 4068289]
    for n in sizes:
        w = np.ones(n)
        for seed in range(12 if n < 10 ** 6 else 3):
            a, b = np.random.RandomState(seed), np.random.RandomState(seed)
            assert _uniform_choice(a, n) == b.choice(n, p=w / w.sum())
            assert a.random_sample() == b.random_sample()

    class Fixed:                                           # extreme draws
        def __init__(self, u):
            self.u = u

        def random_sample(self):
            return self.u

    for n in (5, 1000, 65537):
        p = np.ones(n) / np.ones(n).sum()
        cdf = p.cumsum()
        cdf /= cdf[-1]
        for u in (0.0, np.nextafter(1.0, 0.0), 0.5, 1.0 / n, np.nextafter(1.0 / n, 0), cdf[n // 3], np.nextafter(cdf[n // 3], 0)):
            assert _uniform_choice(Fixed(u), n) == cdf.searchsorted(u, side="right"), (n, u)


def test_repeated_sum_jumps_equal_numpy_cumsum():
    """zk_uniform_choice_index no longer walks the n dependent additions of numpy.cumsum: inside a binade every addition of
    the same constant rounds the same way, so it jumps.  The jumped sums must be numpy's, bit for bit -- constants with and
    without exact ties, at every index of small arrays and at random indices of large ones -- and the index must be the
    sequential replay's."""
    from ctypes import byref, c_double, c_int64
    from mtflearn_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(12)
    consts = [1.0 / n for n in (3, 7, 10, 1000, 4097, 65537, 4068289, 16524225)] + [0.1, 1.0, 2.0 ** -20, 1.5 * 2.0 ** -30, 3.0 * 2.0 ** -40,
              float(np.nextafter(0.25, 1)), 1e-300, 1e300 / 3, float(rng.random()), float(rng.random()) * 1e-9]
    for c in consts:
        cs = np.full(70000, c).cumsum()
        for i in list(range(0, 600)) + list(rng.integers(600, 70000, 200)):
            assert lib.zk_repeated_sum_f64(c, int(i) + 1) == cs[i], (c, i)
    for n in (4068289, 16524225, (1 << 22) + 1):                       # the sizes the bench runs: the last sum and random indices
        c = 1.0 / n
        cs = np.full(n, c).cumsum()
        assert lib.zk_repeated_sum_f64(c, n) == cs[-1]
        for i in rng.integers(0, n, 300):
            assert lib.zk_repeated_sum_f64(c, int(i) + 1) == cs[i], (n, i)
        for u in list(rng.random(20)) + [0.0, float(np.nextafter(1.0, 0.0))]:
            a, b, last = c_int64(), c_int64(), c_double()
            assert lib.zk_uniform_choice_index(n, float(u), byref(a)) == 0
            assert lib.zk_uniform_choice_index_sequential(n, float(u), byref(b), byref(last)) == 0
            assert a.value == b.value == (cs / cs[-1]).searchsorted(u, side="right") and last.value == cs[-1]
