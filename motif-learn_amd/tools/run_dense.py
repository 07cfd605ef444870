#!/usr/bin/env python3
"""Launch the dense / maps / batch kernels of the BASELINE configs a few times each (for rocprofv3 passes).

  python3 run_dense.py [--reps 2] [--which strip8,sep12,maps10,batch12,stream12]
"""
import argparse
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--which", default="strip8,sep12,maps10,batch12,stream12,direct20,points8")
    args = ap.parse_args()
    from mtflearn_amd import ZPs, _native, distributed as D
    from mtflearn_amd.synthetic import honeycomb_frame
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    which = args.which.split(",")

    def zps(n_max, size):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return ZPs(n_max, size)

    if "strip8" in which:
        z = zps(8, 32)
        f = torch.from_numpy(honeycomb_frame(2048, seed=0)).to(dev)
        out = D.frame_moments_device(z._device_plan(), f)
        for _ in range(args.reps):
            D.frame_moments_device(z._device_plan(), f, out=out)
        torch.cuda.synchronize()
        del out
    if "sep12" in which:
        z = zps(12, 64)
        f = torch.from_numpy(honeycomb_frame(4096, seed=3)).to(dev)
        out = D.frame_moments_device(z._device_plan(), f)
        for _ in range(args.reps):
            D.frame_moments_device(z._device_plan(), f, out=out)
        torch.cuda.synchronize()
        del out
    if "maps10" in which:
        z = zps(10, 32)
        f = torch.from_numpy(honeycomb_frame(4096, seed=1)).to(dev)
        theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        for _ in range(args.reps + 1):
            r = D.frame_maps_device(z._device_plan(), f, 36, theta=theta)
        torch.cuda.synchronize()
        del r
    if "batch12" in which or "stream12" in which:
        z = zps(12, 64)
        plan = z._device_plan()
        f = torch.from_numpy(honeycomb_frame(4096, seed=3)).to(dev)
        p = f[:248 + 63].unfold(0, 64, 1).unfold(1, 64, 1).reshape(-1, 64, 64).contiguous()   # 1.0 M windows, 16.4 GB
        out = torch.empty((p.shape[0], 91), dtype=torch.float64, device=dev)
        for name, path in (("batch12", _native.PATH_SEPARABLE), ("stream12", _native.PATH_STREAM)):
            if name in which and plan.has_path(0, _native.ZK_F32, path):
                plan.set_path(path)
                for _ in range(args.reps + 1):
                    D.patch_moments_device(plan, p, out=out)
                torch.cuda.synchronize()
        plan.set_path(_native.PATH_AUTO)
    if "direct20" in which:                                             # what ZK_PATH_AUTO runs from n_max 17 (round 4): the plain sum on the matrix cores
        z = zps(20, 40)
        plan = z._device_plan()
        f = torch.from_numpy(honeycomb_frame(2048, seed=0)).to(dev)
        out = D.frame_moments_device(plan, f, row0=256, n_rows=512)
        for _ in range(args.reps):
            D.frame_moments_device(plan, f, row0=256, n_rows=512, out=out)
        torch.cuda.synchronize()
        del out
        p = f.unfold(0, 40, 3).unfold(1, 40, 3).reshape(-1, 40, 40)[:1 << 18].contiguous()
        o = D.patch_moments_device(plan, p)
        for _ in range(args.reps):
            D.patch_moments_device(plan, p, out=o)
        torch.cuda.synchronize()
        del o, p
    if "points8" in which:                                              # moments at 2^20 random key points (bucketed, 16-byte row loads)
        from ctypes import c_void_p
        z = zps(8, 32)
        plan = z._device_plan()
        f = torch.from_numpy(honeycomb_frame(2048, seed=1)).to(dev)
        pts = torch.from_numpy(np.random.default_rng(0).integers(16, 2032, size=(1 << 20, 2)).astype(np.int32)).to(dev)
        o = torch.empty((1 << 20, 45), dtype=torch.float64, device=dev)
        for _ in range(args.reps + 1):
            _native.check(plan._lib.zk_transform_points_dev(plan._h, c_void_p(f.data_ptr()), 0, 2048, 2048, c_void_p(pts.data_ptr()), 1 << 20,
                                                            c_void_p(o.data_ptr()), None), "points")
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
