#!/usr/bin/env python3
"""Per-launch times of the symmetry maps above n_max 16 (moments into scratch + planes kernel), by requested output.
time_maps_tail.py [n_max size [rows]]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import numpy as np
import torch
from mtflearn_amd import ZPs, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame

n_max, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (28, 56)
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 128
torch.cuda.set_device(0)
theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
f = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    z = ZPs(n_max, K)
plan = z._device_plan()
nc = len(z.n) - sum(1 for m in z.m if m < 0)
for label, kw in (("all", dict(theta=theta)), ("rot only", dict(want_abs=False)), ("abs only", dict(folds=None)),
                  ("mirror only", dict(folds=None, want_abs=False, theta=theta)),
                  ("mirror 48 angles", dict(folds=None, want_abs=False, theta=np.linspace(0, 2 * np.pi, 48, endpoint=False))),
                  ("mirror, irregular 90 angles", dict(folds=None, want_abs=False, theta=np.linspace(0, np.pi, 90)))):
    D.frame_maps_device(plan, f, nc, row0=min(256, 2048 - rows), n_rows=rows, **kw)
    torch.cuda.synchronize()
    plan.profile(True)
    D.frame_maps_device(plan, f, nc, row0=min(256, 2048 - rows), n_rows=rows, **kw)
    torch.cuda.synchronize()
    ms = plan.profile_read_launches()
    plan.profile(False)
    print(f"n_max {n_max} K {K} rows {rows} x 2048  {label:28s}: launches (ms) {[round(x, 3) for x in ms]}", flush=True)
