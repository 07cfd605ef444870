#!/usr/bin/env python3
"""Launch the mixture-moment pass (zk_gmm_moments, k components in one call) and the E step a few times on a resident
4 068 289 x 45 matrix -- for rocprofv3 counter passes.   run_moments.py [k] [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd.clustering import DeviceRows
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, D = 4068289, 45
rng = np.random.default_rng(0)
centres = rng.standard_normal((k, D)) * 2.0
X = centres[rng.integers(0, k, N)] + rng.standard_normal((N, D))
rows = DeviceRows(X)
prec = np.tile(np.eye(D), (k, 1, 1)); logdet = np.zeros(k); logw = np.full(k, -np.log(k))
rows.estep(prec, centres, logdet, logw)
mean = X[:1000].mean(0)
rows.profile(True)
for _ in range(reps):
    rows.moments(0, mean, count=k)
    print("moments kernel ms", rows.last_kernel_ms())
for _ in range(reps):
    rows.estep(prec, centres, logdet, logw)
