#!/usr/bin/env python3
"""Soak of round 4's changes, seeded random problems on the GPU against the CPU oracle (test infrastructure; never the product path):
  (a) what ZK_PATH_AUTO runs (polynomial kernels up to n_max 16, matrix-core plain sum above) and every forced family, batch and dense, by the
      criterion of SURVEY 8(c) with the floors of tests/test_gpu_parity.py::_floor -- dense against the reference's convolution form
      (oracle convolution_basis), NaN pixels outside the disk included;
  (b) key points: bucket order / 16-byte row loads / caller order / 4-byte loads give the same bits, random frames, window sizes, point clouds
      (sparse, dense, clustered, outside the frame) and a sample against the oracle;
  (c) the opt-in strip kernel (ZK_STRIP_V3=1) equals the default one bit for bit.

    soak_round4.py [seed] [iterations]
"""
import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "motif-learn_amd")); sys.path.insert(0, R)
import numpy as np
from mtflearn_amd import ZPs, _native as native
from oracle import zernike_oracle as zo


def zps(n, k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ZPs(n, k)


def floor(name, n_max):
    if name in ("separable", "stream") and n_max > 16:
        return 3e-7 if n_max > 20 else 1e-8
    return 1e-11 if n_max > 12 else 1e-12


def worst(got, ref, fl):
    return float((np.abs(got - ref) / (1e-6 * np.abs(ref) + fl * np.abs(ref).max())).max())


def families(z, arr, mode):
    plan = z._device_plan()
    run = plan.transform_patches if mode == 0 else plan.transform_frame
    out = {"auto": run(arr)}
    for path, name in native.PATH_NAMES.items():
        if plan.has_path(mode, native.dtype_code(arr.dtype), path):
            plan.set_path(path)
            out[name] = run(arr)
    plan.set_path(native.PATH_AUTO)
    return out


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rng = np.random.default_rng(seed)
    bad = n_cmp = 0
    for it in range(iters):
        size = int(rng.integers(8, 73))
        n_max = int(min(size, rng.integers(0, 41 if size >= 16 else 13)))
        dtype = np.float32 if rng.random() < 0.7 else np.float64
        z = zps(n_max, size)
        n, V = z.n, z.polynomials
        # structured + noisy input: a smooth blob pattern plus noise (most moments small against the largest)
        yy, xx = np.mgrid[:size, :size]
        blob = np.exp(-((xx - size * rng.random()) ** 2 + (yy - size * rng.random()) ** 2) / (2 * (size / 8) ** 2))
        p = (blob[None] * rng.random((int(rng.integers(1, 300)), 1, 1)) + 0.05 * rng.random((1, size, size))).astype(dtype)
        img = (rng.random((int(rng.integers(size, size + 40)), int(rng.integers(size, size + 150)))) - 0.3).astype(dtype)
        refs = {0: zo.moments_patches(p, V), 1: zo.moments_frame_direct(img, zo.convolution_basis(V, n))}
        for mode, arr in ((0, p), (1, img)):
            for name, got in families(z, arr, mode).items():
                n_cmp += 1
                w = worst(got, refs[mode], floor(name, n_max))
                if not w <= 1.0:
                    bad += 1
                    print("MISMATCH", size, n_max, dtype.__name__, "batch" if mode == 0 else "dense", name, w, flush=True)
        # a NaN outside the disk of every window never reaches a moment (batch mode, corner pixel)
        if size >= 8 and V[0, 0, 0] == 0.0:
            q = p.copy()
            q[:, 0, 0] = np.nan
            got = z.transform(q).data
            n_cmp += 1
            if not np.array_equal(got, z.transform(p).data):
                bad += 1
                print("NaN outside the disk changed the moments", size, n_max, flush=True)
        # (b) key points
        if n_max <= 16 and it % 2 == 0:
            import torch
            from ctypes import c_void_p
            H, W = int(rng.integers(size + 4, 400)), int(rng.integers(size + 4, 1300))
            frame = rng.random((H, W)).astype(np.float32)
            npts = int(rng.choice([300, 5000, 40000]))
            kind = rng.integers(0, 3)
            pts = np.column_stack([rng.integers(-10, W + 10, npts), rng.integers(-10, H + 10, npts)]).astype(np.int32)
            if kind == 1:
                pts[:, 0] = W // 2 + pts[:, 0] % 5
                pts[:, 1] = H // 2 + pts[:, 1] % 2
            if kind == 2:
                pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
            plan = z._device_plan()
            d_img, d_pts = torch.from_numpy(frame).cuda(), torch.from_numpy(pts).cuda()
            outs = []
            for env in ({}, {"ZK_POINTS_NO_BUCKET": "1"}, {"ZK_POINTS_NO_WIDE": "1"}, {"ZK_POINTS_NO_BUCKET": "1", "ZK_POINTS_NO_WIDE": "1"}):
                for k in ("ZK_POINTS_NO_BUCKET", "ZK_POINTS_NO_WIDE"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                o = torch.full((npts, len(z.n)), float("nan"), dtype=torch.float64, device="cuda")
                native.check(plan._lib.zk_transform_points_dev(plan._h, c_void_p(d_img.data_ptr()), native.ZK_F32, H, W, c_void_p(d_pts.data_ptr()), npts,
                                                               c_void_p(o.data_ptr()), None), "points")
                torch.cuda.synchronize()
                outs.append(o.cpu().numpy())
            for k in ("ZK_POINTS_NO_BUCKET", "ZK_POINTS_NO_WIDE"):
                os.environ.pop(k, None)
            n_cmp += 3
            for o in outs[1:]:
                if not np.array_equal(o, outs[0]):
                    bad += 1
                    print("key points: forms differ", size, n_max, H, W, npts, kind, flush=True)
            pick = rng.choice(npts, min(100, npts), replace=False)
            s1, s2, o_ = size // 2, size - size // 2, size + 12
            padded = np.pad(frame, o_)
            win = np.array([padded[y + o_ - s1:y + o_ + s2, x + o_ - s1:x + o_ + s2] for x, y in pts[pick]])
            n_cmp += 1
            w = worst(outs[0][pick], zo.moments_patches(win, V), floor("auto", n_max))
            if not w <= 1.0:
                bad += 1
                print("key points vs oracle", size, n_max, w, flush=True)
        # (c) strip3 opt-in
        if n_max <= 8 and size % 2 == 0 and size <= 32:
            plan = z._device_plan()
            plan.set_path(native.PATH_SEPARABLE)
            a = plan.transform_frame(img)
            os.environ["ZK_STRIP_V3"] = "1"
            b = plan.transform_frame(img)
            os.environ.pop("ZK_STRIP_V3")
            plan.set_path(native.PATH_AUTO)
            n_cmp += 1
            if not np.array_equal(a, b):
                bad += 1
                print("strip3 differs from strip2", size, n_max, flush=True)
        if it % 10 == 9:
            print("iter", it + 1, "comparisons", n_cmp, "bad", bad, flush=True)
    print("done: seed", seed, "iterations", iters, "comparisons", n_cmp, "bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
