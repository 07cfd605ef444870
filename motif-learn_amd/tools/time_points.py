#!/usr/bin/env python3
"""Times zk_transform_points_dev: N random key points on a resident frame (moments, no patch batch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import torch
from mtflearn_amd import ZPs, _native
from mtflearn_amd.synthetic import honeycomb_frame
from ctypes import c_void_p

n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 8
z = ZPs(n_max, 32); plan = z._device_plan()
img = torch.from_numpy(honeycomb_frame(2048, seed=1)).cuda()
rng = np.random.default_rng(0)
for n_pts, kind in [(1 << 20, "random"), (1 << 20, "sorted"), (100000, "random")]:
    pts = rng.integers(16, 2048 - 16, size=(n_pts, 2)).astype(np.int32)
    if kind == "sorted":
        pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    d_pts = torch.from_numpy(pts).cuda()
    out = torch.empty((n_pts, len(z.n)), dtype=torch.float64, device="cuda")
    lib = plan._lib
    run = lambda: _native.check(lib.zk_transform_points_dev(plan._h, c_void_p(img.data_ptr()), 0, 2048, 2048,
                                                            c_void_p(d_pts.data_ptr()), n_pts, c_void_p(out.data_ptr()), None), "pts")
    run(); torch.cuda.synchronize(); plan.profile(True)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    k, ms = plan.profile_read(); plan.profile(False)
    print(f"{n_pts:8d} {kind:6s} points: {ms / k:7.3f} ms  {n_pts / (ms / k) / 1e3:8.1f} M points/s")
