#!/usr/bin/env python3
"""kNN search at several sizes in one process (for a rocprofv3 --kernel-trace run: the kernel durations, in launch order).
Usage: time_knn_sizes.py N1 N2 ... [--d D] [--k K]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd.clustering import DeviceRows
from mtflearn_amd.manifold import _knn_affinities
args = sys.argv[1:]
D = int(args[args.index("--d") + 1]) if "--d" in args else 45
k = int(args[args.index("--k") + 1]) if "--k" in args else 10
sizes = [int(a) for a in args if a.isdigit() and args[max(0, args.index(a) - 1)] not in ("--d", "--k")]
rng = np.random.default_rng(0)
centres = rng.standard_normal((8, D)) * 2
for N in sizes:
    X = centres[rng.integers(0, 8, N)] + rng.standard_normal((N, D))
    with DeviceRows(X) as rows:
        _knn_affinities(rows, k, 1, k)
        t0 = time.perf_counter(); _knn_affinities(rows, k, 1, k); t = time.perf_counter() - t0
    print(f"N = {N}, D = {D}, k = {k}: {t * 1e3:.1f} ms end to end (second call)")
