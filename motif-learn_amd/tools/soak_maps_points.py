#!/usr/bin/env python3
"""Soak: seeded random plans; ZPs.symmetry_maps (fused / planes kernels) against the container's NumPy methods on the
device moments, and ZPs.transform_at (key-point kernel / device gather) against the batch path on host-cut windows.

  python motif-learn_amd/tools/soak_maps_points.py [seed] [iterations] [n_max below this bound, default 25]
"""
import os, sys, warnings
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "motif-learn_amd")); sys.path.insert(0, R)
from mtflearn_amd import ZPs

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = n_cmp = 0


def check(tag, got, ref, tol):
    global bad, n_cmp
    n_cmp += 1
    scale = np.nanmax(np.abs(ref)) or 1.0
    err = np.nanmax(np.abs(got - ref)) / scale
    if not err <= tol or np.isnan(got).sum() != np.isnan(ref).sum():
        bad += 1
        print("MISMATCH", tag, err, flush=True)


for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    size = int(rng.integers(8, 73))
    n_max = int(min(size, rng.integers(2, int(sys.argv[3]) if len(sys.argv) > 3 else 25)))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z = ZPs(n_max, size)
    frame = (rng.random((int(rng.integers(size, size + 50)), int(rng.integers(size, size + 150)))) + 0.1).astype(dtype)
    zm = z.transform(frame)
    folds = [int(f) for f in rng.choice([1, 2, 3, 4, 5, 6, 8], size=int(rng.integers(1, 5)), replace=False)]
    unsel = (0, 1) if rng.random() < 0.7 else (0, 1, 2)
    theta = None if rng.random() < 0.6 else np.sort(rng.uniform(0, 2 * np.pi, int(rng.integers(1, 50))))
    p = 2 if rng.random() < 0.8 else None
    got = z.symmetry_maps(frame, n_folds=folds, p=p, m_unselect=unsel, theta=theta)
    tag = (size, n_max, dtype.__name__, folds, unsel, p, None if theta is None else len(theta))
    check(("rot",) + tag, got["rot_maps"], zm.rot_maps(folds, p=p, m_unselect=unsel), 1e-9)
    check(("abs",) + tag, got["abs"], np.abs(zm.to_complex().data), 1e-9)
    check(("mirror",) + tag, got["mirror_map"], zm.mirror_map(theta=theta, p=p, m_unselect=unsel), 1e-9)
    H, W = frame.shape
    pts = np.column_stack([rng.integers(-3, W + 3, 40), rng.integers(-3, H + 3, 40)])
    padded = np.pad(frame, size + 4)
    s1 = size // 2
    wins = np.array([padded[y + size + 4 - s1:y + 2 * size + 4 - s1, x + size + 4 - s1:x + 2 * size + 4 - s1] for x, y in pts])
    floor = 3e-7 if n_max > 20 else 1e-8 if n_max > 16 else 1e-10 if n_max > 12 else 1e-11
    plan = z._device_plan()
    plan.set_path(1)                                   # generic kernel on the host-cut windows
    ref_pts = plan.transform_patches(np.ascontiguousarray(wins))
    plan.set_path(0)
    check(("points",) + tag[:3], z.transform_at(frame, pts).data, ref_pts, floor)
    if it % 25 == 24:
        print("iter", it + 1, "comparisons", n_cmp, "bad", bad, flush=True)
print("done comparisons", n_cmp, "bad", bad)
