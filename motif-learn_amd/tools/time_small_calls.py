#!/usr/bin/env python3
"""Latency of small host calls (ZPs.transform on a handful of patches / a small frame): what a call costs before its size matters."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
from mtflearn_amd import ZPs
warnings.simplefilter("ignore")
z = ZPs(8, 32)
rng = np.random.default_rng(0)
for n in (1, 64, 1000, 10000, 100000):
    p = rng.random((n, 32, 32), dtype=np.float32)
    z.transform(p)
    t0 = time.perf_counter()
    for _ in range(10): z.transform(p)
    print(f"{n:7d} patches: {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms per call")
for side in (64, 256, 1024):
    f = rng.random((side, side), dtype=np.float32)
    z.transform(f)
    t0 = time.perf_counter()
    for _ in range(10): z.transform(f)
    print(f"{side:4d}^2 frame: {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms per call")
f = rng.random((512, 512), dtype=np.float32)
z10 = ZPs(10, 32)
z10.symmetry_maps(f)
t0 = time.perf_counter()
for _ in range(10): z10.symmetry_maps(f)
print(f"symmetry_maps 512^2: {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms per call")
