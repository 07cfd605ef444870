#!/usr/bin/env python3
"""Does the batch kernel's time depend on where its buffers land?  Re-allocates the input / output with
different paddings in front (fresh process each call gives yet another physical layout) and times the
same kernel; prints one line per layout."""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import torch
from mtflearn_amd import ZPs
from mtflearn_amd.synthetic import honeycomb_frame
from mtflearn_amd.distributed import patch_moments_device

z = ZPs(8, 32); plan = z._device_plan()
f = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
base = f.unfold(0, 32, 1).unfold(1, 32, 1).reshape(-1, 32, 32).contiguous()
n = base.shape[0]
plan.profile(True)
for trial in range(10):
    torch.cuda.empty_cache()
    pad_in = torch.empty((trial * 37 + 1) * 1024 * 1024 // 4 * 3, dtype=torch.float32, device="cuda")
    src = torch.empty_like(base); src.copy_(base)
    pad_out = torch.empty((trial * 53 + 3) * 1024 * 1024 // 8, dtype=torch.float64, device="cuda")
    out = torch.empty((n, 45), dtype=torch.float64, device="cuda")
    patch_moments_device(plan, src, out=out); torch.cuda.synchronize(); plan.profile_read()
    ts = []
    for _ in range(9):
        patch_moments_device(plan, src, out=out); torch.cuda.synchronize()
        k, ms = plan.profile_read(); ts.append(ms / k)
    print(f"trial {trial}: src@{src.data_ptr() % (1 << 32):#012x} out@{out.data_ptr() % (1 << 32):#012x}  "
          f"median {statistics.median(ts):.3f} ms  min {min(ts):.3f}  max {max(ts):.3f}", flush=True)
    del src, out, pad_in, pad_out
