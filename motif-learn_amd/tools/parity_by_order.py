#!/usr/bin/env python3
"""Every kernel family against REFERENCE outputs on structured inputs (tests/golden: st_* arrays, orders 10 .. 24), by
SURVEY 8c's criterion verbatim -- worst element of |got - ref| / (1e-6 |ref| + floor max|ref|) for floor 1e-12 and 1e-11
(<= 1 passes) -- next to each family's throughput on a resident batch / frame.  This is the measurement behind
ZK_PATH_AUTO's choice of family per order (csrc/zk_api.hip: zk_plan_auto_direct).   parity_by_order.py [--no-time]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import numpy as np
import torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame

ORDERS = ((10, 32), (12, 64), (14, 32), (16, 32), (18, 40), (20, 40), (22, 48), (24, 48))
TIME = "--no-time" not in sys.argv


def sample_index(n, step):
    return np.array(sorted(set(range(0, n, step)) | {n - 1}), dtype=np.int64)


def worst(got, ref, floor):
    return float((np.abs(got - ref) / (1e-6 * np.abs(ref) + floor * np.abs(ref).max())).max())


def main():
    torch.cuda.set_device(0)
    with np.load(os.path.join(ROOT, "tests", "golden", "zps_golden.npz")) as f:
        g = {k: f[k] for k in f.files}
    frame = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
    print("family: worst element vs the reference by floor 1e-12 / 1e-11 (<= 1 passes), max|err| / max|Z|; throughput")
    for n_max, K in ORDERS:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            z = ZPs(n_max, K)
        plan = z._device_plan()
        tag = f"{n_max}_{K}"
        batch, zref = np.ascontiguousarray(g[f"st_batch_{tag}"]), g[f"st_Z_{tag}"]
        # a batch of >= 64 patches so that every family (the direct kernel included) is offered: the goldens repeated
        reps = -(-64 // len(batch))
        big = np.ascontiguousarray(np.concatenate([batch] * reps))
        crop = g[f"st_frame_{tag}"].astype(np.float64)
        H, W = crop.shape
        ri, ci = sample_index(H, 4), sample_index(W, 5)
        fref = g[f"st_Zf_{tag}"]
        auto = [_native.PATH_NAMES[plan.best_path(0, _native.ZK_F32, 1 << 20)], _native.PATH_NAMES[plan.best_path(1, _native.ZK_F64)]]
        print(f"--- n_max {n_max} K {K} ({len(z.n)} moments)   ZK_PATH_AUTO: batch {auto[0]}, dense {auto[1]}")
        n_t = 1 << 18 if K <= 48 else 1 << 17
        pt = frame.unfold(0, K, 3).unfold(1, K, 3).reshape(-1, K, K)[:n_t].contiguous()
        assert pt.shape[0] == n_t, pt.shape  # (a stride of 5 px leaves 161 604 windows of 40 px on a 2048^2 frame: fewer than 2^18)
        for path, name in _native.PATH_NAMES.items():
            line = f"  {name:10s}"
            if plan.has_path(0, _native.ZK_F32, path):
                plan.set_path(path)
                got = plan.transform_patches(big)[:len(batch)]
                line += f" batch: {worst(got, zref, 1e-12):9.3g} / {worst(got, zref, 1e-11):9.3g}  err {np.abs(got - zref).max() / np.abs(zref).max():8.1e}"
                if TIME and name != "generic":
                    out = D.patch_moments_device(plan, pt)
                    torch.cuda.synchronize()
                    plan.profile(True)
                    for _ in range(3):
                        D.patch_moments_device(plan, pt, out=out)
                    torch.cuda.synchronize()
                    _, ms = plan.profile_read()
                    plan.profile(False)
                    line += f"  {n_t / (ms / 3) / 1e3:8.1f} M patches/s"
                    del out
            else:
                line += " batch: -" + " " * 62
            if plan.has_path(1, _native.ZK_F64, path):
                plan.set_path(path)
                got = plan.transform_frame(crop)[:, ri][:, :, ci]
                line += f" | dense: {worst(got, fref, 1e-12):9.3g} / {worst(got, fref, 1e-11):9.3g}  err {np.abs(got - fref).max() / np.abs(fref).max():8.1e}"
                if TIME and name != "generic" and plan.has_path(1, _native.ZK_F32, path):
                    band = 256
                    o2 = D.frame_moments_device(plan, frame, row0=384, n_rows=band)
                    torch.cuda.synchronize()
                    plan.profile(True)
                    for _ in range(3):
                        D.frame_moments_device(plan, frame, row0=384, n_rows=band, out=o2)
                    torch.cuda.synchronize()
                    _, ms = plan.profile_read()
                    plan.profile(False)
                    line += f"  {band * 2048 / (ms / 3) / 1e3:8.1f} M positions/s"
                    del o2
            plan.set_path(_native.PATH_AUTO)
            print(line, flush=True)
        del pt


if __name__ == "__main__":
    main()
