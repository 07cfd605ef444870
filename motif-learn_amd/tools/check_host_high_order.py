#!/usr/bin/env python3
"""ZPs.transform on a NumPy batch large enough for several chunks of the host pipeline (H2D / kernel / D2H on three streams) at an order
the matrix-core kernel serves: every row against the oracle's plain sum on a sample, and the whole result against a second call.
check_host_high_order.py [n_patches]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd")); sys.path.insert(0, ROOT)
import numpy as np
from mtflearn_amd import ZPs
from oracle import zernike_oracle as zo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    z = ZPs(20, 40)
rng = np.random.default_rng(3)
p = rng.random((n, 40, 40), dtype=np.float32)
t0 = time.perf_counter(); a = z.transform(p).data.copy(); t1 = time.perf_counter(); b = z.transform(p).data; t2 = time.perf_counter()
idx = np.r_[0:64, n // 2 - 32:n // 2 + 32, n - 64:n, rng.integers(0, n, 200)]
ref = zo.moments_patches(p[idx], z.polynomials)
err = np.abs(a[idx] - ref).max() / np.abs(ref).max()
print(f"{n} x 40 x 40 float32 from NumPy at n_max 20: {n / (t2 - t1) / 1e6:.2f} M patches/s (second call; first {n / (t1 - t0) / 1e6:.2f}), "
      f"{p.nbytes / (t2 - t1) / 1e9:.1f} GB/s in; both calls identical: {np.array_equal(a, b)}; sample of {len(idx)} rows vs the oracle: {err:.1e} of max|Z|")
assert np.array_equal(a, b) and err < 1e-12
