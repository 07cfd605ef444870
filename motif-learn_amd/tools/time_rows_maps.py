#!/usr/bin/env python3
"""Times zk_moment_maps_dev (the zmoments tail on rank-2 data) on a resident (N, n_poly) moment matrix."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import torch
from ctypes import POINTER, c_double, c_int32, c_void_p
from mtflearn_amd import ZPs, _native

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4068289
for n_max in (8, 12):
    z = ZPs(n_max, 32 if n_max == 8 else 64); plan = z._device_plan(); lib = plan._lib
    n_poly = len(z.n); n_c = sum(n // 2 + 1 for n in range(n_max + 1))
    mom = torch.randn((N, n_poly), dtype=torch.float64, device="cuda")
    rot = torch.empty((N, 4), dtype=torch.float64, device="cuda"); ab = torch.empty((N, n_c), dtype=torch.float64, device="cuda")
    mir = torch.empty((N,), dtype=torch.float64, device="cuda")
    folds = np.array([2, 3, 4, 6], dtype=np.int32); unsel = np.array([0, 1], dtype=np.int32)
    theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
    def run(r, a, m):
        _native.check(lib.zk_moment_maps_dev(plan._h, c_void_p(mom.data_ptr()), N, folds.ctypes.data_as(POINTER(c_int32)), 4,
                                             unsel.ctypes.data_as(POINTER(c_int32)), 2, 2, theta.ctypes.data_as(POINTER(c_double)), 360,
                                             c_void_p(r), c_void_p(a), c_void_p(m), None), "zk_moment_maps_dev")
    for label, args in (("rot + abs + mirror", (rot.data_ptr(), ab.data_ptr(), mir.data_ptr())), ("rot only", (rot.data_ptr(), None, None)),
                        ("abs only", (None, ab.data_ptr(), None))):
        run(*args); torch.cuda.synchronize(); plan.profile(True)
        for _ in range(5):
            run(*args)
        torch.cuda.synchronize(); k, ms = plan.profile_read(); plan.profile(False)
        print(f"n_max {n_max:2d} {label:20s}: {ms / k:7.3f} ms for {N} rows  (input {N * n_poly * 8 / (ms / k) / 1e9:6.2f} TB/s)")
