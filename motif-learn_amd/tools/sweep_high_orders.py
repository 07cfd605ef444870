#!/usr/bin/env python3
"""Throughput of the orders the reference's own estimator picks for large patches (n_max = size / 2: 28 / 32 / 36 at 56 / 64 /
72 px) -- the generic kernel's territory (no fold or polynomial substitution is exact there) -- next to
n_max 24, the last order of the separable kernels.   sweep_high_orders.py"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import numpy as np
import torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame

torch.cuda.set_device(0)
theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
f = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
for n_max, K in ((24, 56), (28, 56), (32, 64), (36, 72)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z = ZPs(n_max, K)
    plan = z._device_plan()
    n = 1 << 17
    p = f.unfold(0, K, 5).unfold(1, K, 5).reshape(-1, K, K)[:n].contiguous()
    out = D.patch_moments_device(plan, p)
    torch.cuda.synchronize()
    plan.profile(True)
    for _ in range(3):
        D.patch_moments_device(plan, p, out=out)
    torch.cuda.synchronize()
    k, ms = plan.profile_read()
    ms_b = ms / 3
    band = 256
    o2 = D.frame_moments_device(plan, f, row0=384, n_rows=band)
    torch.cuda.synchronize()
    plan.profile_read()
    D.frame_moments_device(plan, f, row0=384, n_rows=band, out=o2)
    torch.cuda.synchronize()
    k, ms_f = plan.profile_read()
    # the symmetry maps of a 512-row band (moments from the dense kernel into scratch + the planes kernel)
    mrows = 512
    maps = D.frame_maps_device(plan, f, len(z.n) - sum(1 for m in z.m if m < 0), row0=256, n_rows=mrows, theta=theta)
    torch.cuda.synchronize()
    plan.profile_read()
    D.frame_maps_device(plan, f, len(z.n) - sum(1 for m in z.m if m < 0), row0=256, n_rows=mrows, theta=theta)
    torch.cuda.synchronize()
    k_m, ms_m = plan.profile_read()
    plan.profile(False)
    kern = _native.PATH_NAMES[plan.best_path(0, _native.ZK_F32)]
    print(f"n_max {n_max:2d} K {K:3d} ({len(z.n)} moments)  batch [{kern}]: {p.shape[0] / ms_b / 1e3:8.2f} M patches/s   "
          f"dense [{_native.PATH_NAMES[plan.best_path(1, _native.ZK_F32)]}]: {band * 2048 / ms_f / 1e3:8.2f} M positions/s   "
          f"maps (rot 4 folds, |Z|, mirror 360): {mrows * 2048 / ms_m / 1e3:7.2f} M positions/s ({k_m} launches)", flush=True)
    del out, o2, p, maps
