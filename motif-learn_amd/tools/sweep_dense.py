#!/usr/bin/env python3
"""Dense-kernel sweep: kernel ms per frame for a list of (size, n_max) on one synthetic frame, the library's
HIP-event profiler, ZK_PATH_SEPARABLE (strip form where it exists unless ZK_NO_STRIP is set) and the fused maps.

  python motif-learn_amd/tools/sweep_dense.py [--frame 2048] [--cases 32:8,32:10,64:12,...] [--maps]
Run once per library build (MTFLEARN_AMD_LIB=...) to A/B a compile-time switch."""
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
    ap.add_argument("--frame", type=int, default=2048)
    ap.add_argument("--cases", default="32:6,32:8,32:10,32:12,48:10,64:10,64:12,64:14,64:16,72:12,96:16,48:20")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--maps", action="store_true")
    args = ap.parse_args()
    from mtflearn_amd import ZPs, _native, distributed as D
    from mtflearn_amd.synthetic import honeycomb_frame
    torch.cuda.set_device(0)
    f = torch.from_numpy(honeycomb_frame(args.frame, seed=0)).cuda()
    theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
    print("lib", _native.LIB_PATH)
    for case in args.cases.split(","):
        K, n_max = (int(x) for x in case.split(":"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            z = ZPs(n_max, K)
        plan = z._device_plan()
        out = D.frame_moments_device(plan, f)
        torch.cuda.synchronize()
        plan.profile(True)
        for _ in range(args.reps):
            D.frame_moments_device(plan, f, out=out)
        torch.cuda.synchronize()
        _, ms = plan.profile_read()
        line = f"K={K:3d} n_max={n_max:2d}  dense {ms / args.reps:8.3f} ms"
        del out
        if args.maps and plan.supports(_native.OP_MAPS, _native.ZK_F32):
            n_c = sum(n // 2 + 1 for n in range(n_max + 1))
            for label, kw in (("maps(all)", dict(theta=theta)), ("maps(rot+abs)", dict(theta=None))):
                r = D.frame_maps_device(plan, f, n_c, **kw)
                torch.cuda.synchronize()
                plan.profile_read()
                for _ in range(args.reps):
                    D.frame_maps_device(plan, f, n_c, **kw)
                torch.cuda.synchronize()
                _, ms = plan.profile_read()
                line += f"  {label} {ms / args.reps:8.3f} ms"
                del r
        plan.profile(False)
        print(line, flush=True)


if __name__ == "__main__":
    main()
