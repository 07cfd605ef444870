#!/usr/bin/env python3
"""Where the wall-clock time of gmm_fit_predict on a resident 4 M x 45 matrix goes (cProfile by function)."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd.clustering import DeviceRows, gmm_fit_predict
N, D, k = 4068289, 45, int(sys.argv[1]) if len(sys.argv) > 1 else 6
cov = sys.argv[2] if len(sys.argv) > 2 else "full"
rng = np.random.default_rng(0)
centres = rng.standard_normal((k, D)) * 3
X = centres[rng.integers(0, k, N)] + rng.standard_normal((N, D))
with DeviceRows(X) as rows:
    gmm_fit_predict(rows, k, cov, 0)
    t0 = time.perf_counter(); out = gmm_fit_predict(rows, k, cov, 0); t = time.perf_counter() - t0
    print(f"gmm_fit_predict ({cov}): {t * 1e3:.1f} ms")
    pr = cProfile.Profile(); pr.enable(); gmm_fit_predict(rows, k, cov, 0); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
