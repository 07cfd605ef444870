#!/usr/bin/env python3
"""One batch call at a high order (for rocprofv3 --kernel-trace --stats).   run_high_order.py [n_max] [K] [log2 patches]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import torch
from mtflearn_amd import ZPs, distributed as D
n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 28
K = int(sys.argv[2]) if len(sys.argv) > 2 else 56
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 15)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    z = ZPs(n_max, K)
plan = z._device_plan()
p = torch.rand((n, K, K), device="cuda")
out = D.patch_moments_device(plan, p)
torch.cuda.synchronize()
for _ in range(2):
    D.patch_moments_device(plan, p, out=out)
torch.cuda.synchronize()
print("done", float(out[0, 0]))
