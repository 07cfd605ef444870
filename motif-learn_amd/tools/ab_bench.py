#!/usr/bin/env python3
"""Interleaved A/B timing of batch-kernel builds in ONE process (cdna_hip_programming.md rule 24).

Every argument is a path to a build of libzernike_hip.so (same ABI).  For each library a plan for
(n_max, size) is created; then `rounds` rounds run every variant once per round on the same device
buffers, timed with the library's own HIP-event profiler.  Prints median / min / max kernel ms.

  python motif-learn_amd/tools/ab_bench.py [--frame 2048] [--rounds 15] lib_a.so lib_b.so ...
"""
import argparse
import ctypes
import os
import statistics
import sys
from ctypes import POINTER, byref, c_double, c_int, c_int32, c_int64, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--frame", type=int, default=2048)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--n-max", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--mode", choices=["patches", "frame"], default="patches")
    ap.add_argument("--f64", action="store_true", help="float64 operands")
    ap.add_argument("--path", type=int, default=0, help="ZK_PATH_* to force (0 auto, 1 generic, 2 folded, 3 separable, 4 stream)")
    args = ap.parse_args()

    torch.cuda.set_device(0)
    from mtflearn_amd import ZPs
    from mtflearn_amd.synthetic import honeycomb_frame
    z = ZPs(args.n_max, args.size)
    basis = np.ascontiguousarray(z.polynomials)
    n32, m32 = z.n.astype(np.int32), z.m.astype(np.int32)
    K, H = args.size, args.frame
    f_dev = torch.from_numpy(honeycomb_frame(H, seed=0)).cuda()
    if args.f64:
        f_dev = f_dev.double()
    code, esz = (1, 8) if args.f64 else (0, 4)
    if args.mode == "patches":
        src = f_dev.unfold(0, K, 1).unfold(1, K, 1).reshape(-1, K, K).contiguous()
        n_units = src.shape[0]
        out = torch.empty((n_units, len(z.n)), dtype=torch.float64, device="cuda")
        bytes_per_unit = K * K * esz + 8 * len(z.n)
    else:
        src = f_dev
        n_units = H * H
        out = torch.empty((len(z.n), H, H), dtype=torch.float64, device="cuda")
        bytes_per_unit = esz + 8 * len(z.n)
    torch.cuda.synchronize()

    plans = []
    for path in args.libs:
        lib = ctypes.CDLL(os.path.abspath(path))
        lib.zk_plan_create.argtypes = [c_int, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_double), c_int, POINTER(c_void_p)]
        lib.zk_transform_patches_dev.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]
        lib.zk_transform_frame_dev.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]
        lib.zk_plan_profile.argtypes = [c_void_p, c_int]
        lib.zk_plan_profile_read.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_double)]
        h = c_void_p()
        rc = lib.zk_plan_create(K, len(z.n), n32.ctypes.data_as(POINTER(c_int32)), m32.ctypes.data_as(POINTER(c_int32)),
                                basis.ctypes.data_as(POINTER(c_double)), 0, byref(h))
        assert rc == 0, (path, rc)
        lib.zk_plan_set_path.argtypes = [c_void_p, c_int]
        lib.zk_plan_set_path(h, args.path)
        lib.zk_plan_profile(h, 1)
        plans.append((os.path.basename(path), lib, h, []))

    def run(lib, h):
        if args.mode == "patches":
            rc = lib.zk_transform_patches_dev(h, c_void_p(src.data_ptr()), code, n_units, c_void_p(out.data_ptr()), None)
        else:
            rc = lib.zk_transform_frame_dev(h, c_void_p(src.data_ptr()), code, H, H, 0, H, c_void_p(out.data_ptr()), None)
        assert rc == 0

    for name, lib, h, ts in plans:          # warm-up
        run(lib, h)
    torch.cuda.synchronize()
    for name, lib, h, ts in plans:
        lib.zk_plan_profile_read(h, None, None)
    for r in range(args.rounds):
        order = plans if r % 2 == 0 else plans[::-1]
        for name, lib, h, ts in order:
            run(lib, h)
            n, ms = c_int64(), c_double()
            lib.zk_plan_profile_read(h, byref(n), byref(ms))
            ts.append(ms.value)  # one call per read: the sum over its launches (class-pass kernels: several)
    for name, lib, h, ts in plans:
        med = statistics.median(ts)
        print(f"{name:36s} median {med:7.3f} ms  min {min(ts):7.3f}  max {max(ts):7.3f}   "
              f"{n_units / med / 1e6:8.1f} M units/s  {n_units * bytes_per_unit / med / 1e6:7.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
