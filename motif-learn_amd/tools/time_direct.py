#!/usr/bin/env python3
"""Throughput of the plain sum on the matrix cores (ZK_PATH_DIRECT: what ZK_PATH_AUTO runs from n_max 17), batch and dense, steady state.
Switch: ZK_DIRECT_CH96=1 (96 functions per chunk for every set, round 3's form).   time_direct.py [--batch-only] [--patches N]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import numpy as np
import torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame

torch.cuda.set_device(0)
frame = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
for n_max, K in ((18, 40), (19, 40), (20, 40), (23, 48), (24, 48), (28, 56), (36, 72)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z = ZPs(n_max, K)
    plan = z._device_plan()
    plan.set_path(_native.PATH_DIRECT)
    n_t = 1 << 18 if K <= 48 else 1 << 17
    if "--patches" in sys.argv:
        n_t = int(sys.argv[sys.argv.index("--patches") + 1])
    pt = frame.unfold(0, K, 3).unfold(1, K, 3).reshape(-1, K, K)[:n_t].contiguous()
    assert pt.shape[0] == n_t, pt.shape      # (a stride of 5 px leaves 161 604 windows of 40 px on a 2048^2 frame: fewer than 2^18)
    out = D.patch_moments_device(plan, pt)
    torch.cuda.synchronize()
    fl = 2.0 * int(np.count_nonzero(z.polynomials[0])) * len(z.n)
    for _ in range(10):
        D.patch_moments_device(plan, pt, out=out)
    plan.profile(True)
    for _ in range(10):
        D.patch_moments_device(plan, pt, out=out)
    torch.cuda.synchronize()
    _, ms = plan.profile_read()
    plan.profile(False)
    line = f"({K}, {n_max}) {len(z.n):3d} functions  batch: {n_t / (ms / 10) / 1e3:7.1f} M patches/s = {n_t * fl / (ms / 10) / 1e9 / 78.6:.2f} of the FP64 peak"
    del out, pt
    if "--batch-only" in sys.argv:
        print(line, flush=True)
        continue
    band = 256
    o2 = D.frame_moments_device(plan, frame, row0=384, n_rows=band)
    for _ in range(10):
        D.frame_moments_device(plan, frame, row0=384, n_rows=band, out=o2)
    plan.profile(True)
    for _ in range(10):
        D.frame_moments_device(plan, frame, row0=384, n_rows=band, out=o2)
    torch.cuda.synchronize()
    _, ms = plan.profile_read()
    plan.profile(False)
    print(line + f"   dense: {band * 2048 / (ms / 10) / 1e3:7.1f} M positions/s = {band * 2048 * fl / (ms / 10) / 1e9 / 78.6:.2f}", flush=True)
    del o2
