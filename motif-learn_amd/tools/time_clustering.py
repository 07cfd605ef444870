#!/usr/bin/env python3
"""Times the clustering passes (csrc/zk_cluster.hip) on a resident (N, D) matrix: per-pass wall time of the synchronous C-ABI
calls (kernel + reduction + a few hundred bytes of D2H) and the whole kmeans_lbs / gmm_lbs against scikit-learn on the host.
Usage: time_clustering.py [N] [D] [k] [--sklearn]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd.clustering import DeviceRows, kmeans_fit, gmm_fit_predict, _row_norms_sq, _Shards

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 4068289
D = int(args[1]) if len(args) > 1 else 45
k = int(args[2]) if len(args) > 2 else 6
rng = np.random.default_rng(0)
centres = rng.standard_normal((k, D)) * 2.0
X = centres[rng.integers(0, k, N)] + rng.standard_normal((N, D))
gb = N * D * 8 / 1e9
t0 = time.perf_counter(); rows = DeviceRows(X); t_up = time.perf_counter() - t0
print(f"matrix {N} x {D} float64 = {gb:.2f} GB, upload {t_up * 1e3:.0f} ms")


def timed(label, fn, reps=5, bytes_=gb):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"  {label:34s} {ms:8.3f} ms   {bytes_ / ms:7.2f} TB/s of the matrix")
    return out


t0 = time.perf_counter(); mean, var, bad = _Shards(rows).center(); print(f"  center (3 passes)                  {(time.perf_counter() - t0) * 1e3:8.3f} ms")
cand = rows.fetch(rng.integers(0, N, 4))
timed("seed step, 4 candidates (first)", lambda: rows.seed_step(cand, _row_norms_sq(cand), False))
rows.seed_pick(0)
timed("seed step, 4 candidates (fold)", lambda: rows.seed_step(cand, _row_norms_sq(cand), True))
timed("seed pick + 4 searches", lambda: rows.seed_pick(1, np.array([1.0, 1e3, 1e5, 1e6])), bytes_=N * 8 / 1e9)
c0 = centres - mean
timed(f"lloyd pass, k = {k}, with sums", lambda: rows.lloyd(c0, True))
timed(f"lloyd pass, k = {k}, labels only", lambda: rows.lloyd(c0, False))
# full upper-triangular factors ('full' / 'tied' covariances) and diagonal ones ('diag' / 'spherical': the matrix-core E step runs
# only the diagonal pieces for those)
prec = np.stack([np.triu(rng.standard_normal((D, D)) * 0.02, 1) + np.eye(D) for _ in range(k)])
prec_diag = np.tile(np.eye(D), (k, 1, 1)); logdet = np.zeros(k); logw = np.full(k, -np.log(k))
timed(f"mixture E step, k = {k}, full factors", lambda: rows.estep(prec, centres, logdet, logw))
timed(f"mixture E step, k = {k}, diagonal", lambda: rows.estep(prec_diag, centres, logdet, logw))
timed("mixture moments, one component", lambda: rows.moments(0, mean))
timed(f"mixture moments, all {k} in one call", lambda: rows.moments(0, mean, count=k))
rows.profile(True)
for label, fn in (("E step, full factors", lambda: rows.estep(prec, centres, logdet, logw)),
                  ("E step, diagonal factors", lambda: rows.estep(prec_diag, centres, logdet, logw)), ("moments, one component", lambda: rows.moments(0, mean)),
                  (f"moments, all {k}", lambda: rows.moments(0, mean, count=k)), ("lloyd with sums", lambda: rows.lloyd(c0, True))):
    fn(); ms = []
    for _ in range(5):
        fn(); ms.append(rows.last_kernel_ms())
    print(f"  kernel only (HIP events): {label:28s} {sorted(ms)[2]:8.3f} ms")
rows.profile(False)
t0 = time.perf_counter(); lab, _, it = kmeans_fit(rows, k, 0); t_km = time.perf_counter() - t0
print(f"kmeans_fit: {t_km * 1e3:.0f} ms, {it} iterations")
t0 = time.perf_counter(); lab2, it2, conv = gmm_fit_predict(rows, k, "full", 0); t_gm = time.perf_counter() - t0
print(f"gmm_fit_predict (full): {t_gm * 1e3:.0f} ms, {it2} EM iterations, converged {conv}")
if "--sklearn" in sys.argv:
    from sklearn.cluster import KMeans
    from sklearn.mixture import GaussianMixture
    n_s = min(N, 500000)
    t0 = time.perf_counter(); m = KMeans(k, random_state=0).fit(X[:n_s]); t = time.perf_counter() - t0
    print(f"sklearn KMeans on {n_s} rows: {t * 1e3:.0f} ms, {m.n_iter_} iterations ({os.cpu_count()} host CPUs)")
    t0 = time.perf_counter(); g = GaussianMixture(k, random_state=0).fit(X[:n_s]); g.predict(X[:n_s]); t = time.perf_counter() - t0
    print(f"sklearn GaussianMixture on {n_s} rows: {t * 1e3:.0f} ms, {g.n_iter_} iterations")
