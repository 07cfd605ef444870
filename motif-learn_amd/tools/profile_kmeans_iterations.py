import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "motif-learn_amd")
from mtflearn_amd.clustering import DeviceRows, kmeans_fit
N, D, k = 4068289, 45, 4
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D))            # no structure: many iterations
with DeviceRows(X) as rows:
    kmeans_fit(rows, k, 0, max_iter=40)
    t0 = time.perf_counter(); _, _, it = kmeans_fit(rows, k, 0, max_iter=40); t = time.perf_counter() - t0
    print(f"kmeans_fit: {t * 1e3:.1f} ms, {it} iterations -> {t / it * 1e3:.3f} ms per iteration")
    pr = cProfile.Profile(); pr.enable(); kmeans_fit(rows, k, 0, max_iter=40); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
