#!/usr/bin/env python3
"""Times the fused symmetry-map kernel with different output subsets (which part of the tail costs what)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import torch
from mtflearn_amd import ZPs, _native
from mtflearn_amd.synthetic import honeycomb_frame
from mtflearn_amd.distributed import frame_maps_device, frame_moments_device

n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 10
side = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
z = ZPs(n_max, 32); plan = z._device_plan()
img = torch.from_numpy(honeycomb_frame(side, seed=1)).cuda()
n_c = sum(n // 2 + 1 for n in range(n_max + 1))
theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
cases = {
    "rot only": dict(folds=(2, 3, 4, 6), want_abs=False, theta=None),
    "abs only": dict(folds=None, want_abs=True, theta=None),
    "mirror only (uniform 360)": dict(folds=None, want_abs=False, theta=theta),
    "mirror only (generic 360)": dict(folds=None, want_abs=False, theta=theta + 1e-3),
    "all": dict(folds=(2, 3, 4, 6), want_abs=True, theta=theta),
}
plan.profile(True)
for name, kw in cases.items():
    frame_maps_device(plan, img, n_c, **kw); torch.cuda.synchronize(); plan.profile_read()
    for _ in range(3):
        frame_maps_device(plan, img, n_c, **kw)
    torch.cuda.synchronize()
    n, ms = plan.profile_read()
    print(f"{name:28s} {ms / n:8.3f} ms")
out = frame_moments_device(plan, img); torch.cuda.synchronize(); plan.profile_read()
for _ in range(3):
    frame_moments_device(plan, img, out=out)
torch.cuda.synchronize()
n, ms = plan.profile_read()
print(f"{'moments only':28s} {ms / n:8.3f} ms")
