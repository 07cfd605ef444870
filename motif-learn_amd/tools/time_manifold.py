#!/usr/bin/env python3
"""Times the manifold consumer: device kNN graph (zk_rows_knn_correlation) against scikit-learn's brute-force search on the host,
and the host optimiser's sweep rate.  Usage: time_manifold.py [N] [D] [k]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd.clustering import DeviceRows
from mtflearn_amd.manifold import _knn_affinities, ForceGraph8

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 45
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(0)
centres = rng.standard_normal((8, D)) * 2
X = centres[rng.integers(0, 8, N)] + rng.standard_normal((N, D))
with DeviceRows(X) as rows:
    _knn_affinities(rows, k, 1, k)
    t0 = time.perf_counter(); dist, ind, P = _knn_affinities(rows, k, 1, k); t = time.perf_counter() - t0
print(f"device kNN + affinities, {N} x {D}, k = {k}: {t * 1e3:.1f} ms  ({N * N * D * 2 / t / 1e12:.2f} TFLOP/s of dot products)")
if "--sklearn" in sys.argv:
    from sklearn.neighbors import NearestNeighbors
    n_s = min(N, 20000)
    t0 = time.perf_counter(); NearestNeighbors(n_neighbors=k, metric="correlation").fit(X[:n_s]).kneighbors(X[:n_s]); t = time.perf_counter() - t0
    print(f"sklearn brute-force correlation kNN on {n_s} rows: {t * 1e3:.0f} ms (x {(N / n_s) ** 2:.0f} for {N} rows)")
n_f = min(N, 20000)
t0 = time.perf_counter(); fg = ForceGraph8(num_iterations=10); fg.fit(X[:n_f]); t = time.perf_counter() - t0
print(f"ForceGraph8.fit on {n_f} rows, 10 sweeps over {len(fg.pairs)} pairs: {t * 1e3:.0f} ms")
