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
for n_pts, kind, bucket, wide in [(1 << 20, "random", True, True), (1 << 20, "random", True, False), (1 << 20, "random", False, True),
                                  (1 << 20, "random", False, False), (1 << 20, "sorted", True, True), (1 << 20, "sorted", False, False),
                                  (100000, "random", True, True), (100000, "random", False, False)]:
    os.environ.pop("ZK_POINTS_NO_BUCKET", None)
    os.environ.pop("ZK_POINTS_NO_WIDE", None)
    if not wide:
        os.environ["ZK_POINTS_NO_WIDE"] = "1"
    if not bucket:
        os.environ["ZK_POINTS_NO_BUCKET"] = "1"
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
    per_call = ms * (k // 5 if k >= 5 else 1) / k      # the bucketing kernels are not profiled launches: HIP events around the moment kernel only
    print(f"{n_pts:8d} {kind:6s} points, {'bucketed' if bucket else 'caller order'}, {'16-B row loads' if wide else '4-B loads'}: moment kernel {ms / k:7.3f} ms  {n_pts / (ms / k) / 1e3:8.1f} M points/s", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"         whole call (default stream, incl. bucketing): {e0.elapsed_time(e1) / 5:7.3f} ms  {n_pts / (e0.elapsed_time(e1) / 5) / 1e3:8.1f} M points/s")
