#!/usr/bin/env python3
"""Batch-kernel sweep over patch sizes: every available kernel family (ZK_PATH_*) on the same device
buffers, interleaved, timed with the library's HIP-event profiler; also checks that the families agree.

  python motif-learn_amd/tools/sweep_batch.py [--gb 6] [--rounds 9] [--cases 24:8:f32,48:10:f32,...]

Prints, per case, kernel ms and algorithmic TB/s (K*K*s + 8*N_poly bytes per patch) for each family.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))

import numpy as np
import torch

DEFAULT = ("16:6:f32,24:8:f32,31:8:f32,32:8:f32,33:8:f32,40:8:f32,48:8:f32,48:10:f32,56:8:f32,64:8:f32,"
           "72:10:f32,72:12:f32,96:10:f32,24:8:f64,33:8:f64,40:10:f64,48:8:f64")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=DEFAULT)
    ap.add_argument("--gb", type=float, default=6.0, help="input bytes per case")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--paths", default="3,4")
    args = ap.parse_args()

    import warnings
    from mtflearn_amd import ZPs, _native
    torch.cuda.set_device(0)
    paths = [int(x) for x in args.paths.split(",")]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for case in args.cases.split(","):
        ks, ns, ds = case.split(":")
        K, n_max = int(ks), int(ns)
        dt = torch.float32 if ds == "f32" else torch.float64
        code, esz = (_native.ZK_F32, 4) if ds == "f32" else (_native.ZK_F64, 8)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            z = ZPs(n_max, K)
        plan = z._device_plan()
        n_poly = len(z.n)
        n = int(args.gb * 1e9 / (K * K * esz)) // 256 * 256 + 37  # ragged tail on purpose
        src = torch.rand((n, K, K), dtype=dt, device="cuda", generator=gen)
        outs = {p: torch.empty((n, n_poly), dtype=torch.float64, device="cuda") for p in paths}
        have = [p for p in paths if plan.has_path(0, code, p)]
        times = {p: [] for p in have}
        for p in have:  # warm-up
            plan.set_path(p)
            plan.transform_patches_dev(src.data_ptr(), code, n, outs[p].data_ptr(), 0)
        torch.cuda.synchronize()
        plan.profile(True)
        for r in range(args.rounds):
            for p in (have if r % 2 == 0 else have[::-1]):
                plan.set_path(p)
                plan.profile_read()
                plan.transform_patches_dev(src.data_ptr(), code, n, outs[p].data_ptr(), 0)
                torch.cuda.synchronize()
                ln, ms = plan.profile_read()
                times[p].append(ms)  # one call per read: the sum over its launches (n_max > 16: 4 passes + 1)
        plan.profile(False)
        plan.set_path(_native.PATH_AUTO)
        bytes_alg = n * (K * K * esz + 8 * n_poly)
        cells = []
        for p in have:
            med = statistics.median(times[p])
            cells.append(f"{_native.PATH_NAMES[p]:>9s} {med:8.3f} ms {bytes_alg / med * 1e-9:6.2f} TB/s")
        scale = float(outs[have[0]].abs().max())
        dev = max(float((outs[p] - outs[have[0]]).abs().max()) for p in have) / scale
        print(f"K={K:4d} n_max={n_max:2d} {ds}  N={n:9d}  " + "  |".join(cells) + f"  | max rel dev {dev:.1e}", flush=True)
        del src, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
