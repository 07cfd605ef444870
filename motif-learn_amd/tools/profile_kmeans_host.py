#!/usr/bin/env python3
"""Where the wall-clock time of kmeans_fit on a resident 4 M x 45 matrix goes (cProfile by function; the kernels are ~2 ms of it)."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd.clustering import DeviceRows, kmeans_fit
N, D, k = 4068289, 45, int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
centres = rng.standard_normal((k, D)) * 3
X = centres[rng.integers(0, k, N)] + rng.standard_normal((N, D))
with DeviceRows(X) as rows:
    kmeans_fit(rows, k, 0)
    t0 = time.perf_counter(); _, _, it = kmeans_fit(rows, k, 0); t = time.perf_counter() - t0
    print(f"kmeans_fit: {t * 1e3:.1f} ms, {it} iterations")
    pr = cProfile.Profile(); pr.enable(); kmeans_fit(rows, k, 0); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
