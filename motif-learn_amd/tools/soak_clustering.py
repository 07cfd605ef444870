#!/usr/bin/env python3
"""Soak of the clustering consumers against scikit-learn: random problems -- separated and overlapping blobs, heavy duplicate
rows (ties, empty clusters -> the relocation path), tiny sample counts, many clusters -- k-means and all four mixture types.
Prints one line per mismatch category and a summary.  Usage: soak_clustering.py [n_cases] [seed]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from sklearn.cluster import KMeans
from sklearn.mixture import GaussianMixture
from mtflearn_amd.clustering import kmeans_fit, gmm_fit_predict

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
stats = {"kmeans": [0, 0, 0, 0], "gmm": [0, 0, 0, 0]}          # cases, exact, n_iter equal, raised alike
worst = 1.0
warnings.simplefilter("ignore")
for case in range(n_cases):
    kind = rng.choice(["blobs", "overlap", "duplicates", "tiny", "many"])
    d = int(rng.choice([2, 3, 7, 16, 45, 66, 91]))
    if kind == "tiny":
        n, k = int(rng.integers(2, 40)), int(rng.integers(1, 6))
    elif kind == "many":
        n, k = int(rng.integers(300, 3000)), int(rng.integers(9, 40))
    else:
        n, k = int(rng.integers(100, 6000)), int(rng.integers(2, 9))
    k = min(k, n)
    true_k = max(1, k + int(rng.integers(-1, 2)))
    spread = {"blobs": 0.3, "overlap": 1.5}.get(kind, 0.8)
    centres = rng.standard_normal((true_k, d)) * 2
    X = centres[rng.integers(0, true_k, n)] + rng.standard_normal((n, d)) * spread
    if kind == "duplicates":
        X = X[rng.integers(0, max(2, n // 50), n)]                   # ~50 copies of each distinct row
    seed = int(rng.integers(0, 100))
    # ---- k-means
    st = stats["kmeans"]; st[0] += 1
    try:
        ref = KMeans(n_clusters=k, random_state=seed).fit(X)
        err_ref = None
    except Exception as e:
        err_ref = type(e).__name__
    try:
        labels, centers, n_iter = kmeans_fit(X, k, random_state=seed)
        err = None
    except Exception as e:
        err = type(e).__name__
    if err_ref or err:
        st[3] += err_ref == err
        if err_ref != err:
            print(f"case {case} kmeans {kind} n={n} d={d} k={k}: sklearn {err_ref} / device {err}")
    else:
        agree = float(np.mean(labels == ref.labels_))
        worst = min(worst, agree)
        st[1] += agree == 1.0
        st[2] += n_iter == ref.n_iter_
        if agree < 1.0 or n_iter != ref.n_iter_:
            print(f"case {case} kmeans {kind} n={n} d={d} k={k} seed={seed}: agreement {agree:.5f}, n_iter {n_iter} vs {ref.n_iter_}")
    # ---- mixture
    if n >= 2 * k and d <= 45 and kind != "duplicates":
        cov = str(rng.choice(["full", "tied", "diag", "spherical"]))
        st = stats["gmm"]; st[0] += 1
        try:
            m = GaussianMixture(k, covariance_type=cov, random_state=seed).fit(X); ref_l = m.predict(X); err_ref = None
        except Exception as e:
            err_ref = type(e).__name__
        try:
            labels, n_iter, conv = gmm_fit_predict(X, k, covariance_type=cov, random_state=seed); err = None
        except Exception as e:
            err = type(e).__name__
        if err_ref or err:
            st[3] += err_ref == err
            if err_ref != err:
                print(f"case {case} gmm {cov} {kind} n={n} d={d} k={k}: sklearn {err_ref} / device {err}")
        else:
            agree = float(np.mean(labels == ref_l))
            st[1] += agree == 1.0
            st[2] += n_iter == m.n_iter_
            if agree < 1.0 or n_iter != m.n_iter_:
                print(f"case {case} gmm {cov} {kind} n={n} d={d} k={k} seed={seed}: agreement {agree:.5f}, n_iter {n_iter} vs {m.n_iter_}")
for name, (cases, exact, iters, raised) in stats.items():
    print(f"{name}: {cases} cases, {exact} with identical labels, {iters} with the same n_iter, {raised} where both raised the same error")
print(f"worst k-means label agreement {worst:.5f}")
