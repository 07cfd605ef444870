#!/usr/bin/env python3
"""Generate tests/golden/zps_high_golden.npz from the REFERENCE implementation (build container only): the orders its own
estimator returns for 56 .. 72-px windows (``features/_estimate_n_max.py:95,123``: up to size / 2), where the build runs the
plain sum on the matrix cores (csrc/zk_direct_patches.hip) and where the reference's float64 basis is neither the exact
polynomial nor point-symmetric any more.

TEST INFRASTRUCTURE, same rules as oracle/make_golden.py (whose loader this uses): reference outputs recorded as plain
arrays, structured inputs (the reference's own test blobs + windows / a crop of a honeycomb lattice), batch path
(``_zps.py:146-157``) and dense path (``_zps.py:159-193``) on a float64-cast crop -- strided positions incl. both zero-padded
borders, per-plane sums over ALL positions.  threadpoolctl pins BLAS to one thread so that the file regenerates bit for bit.

Usage:  python oracle/make_golden_high_orders.py
"""
import os

import numpy as np
from threadpoolctl import threadpool_limits

from make_golden import import_reference, sample_index

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "zps_high_golden.npz")
HIGH_ORDERS = ((28, 56), (32, 64), (36, 72))


def main():
    import warnings
    ZPs, _zm, get_patches, HoneyComb = import_reference()
    lattice = HoneyComb(size=256, l=12, seed=11).to_image()                      # float32 (256, 256)
    g = {}
    with threadpool_limits(1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n_max, K in HIGH_ORDERS:
            z = ZPs(n_max, K)
            tag = f"{n_max}_{K}"
            blobs = np.concatenate([get_patches(size=K, n_fold=3, num_patches=3), get_patches(size=K, n_fold=6, num_patches=2)])
            crops = np.array([lattice[r:r + K, c:c + K] for r in range(7, 256 - K, 83) for c in range(2, 256 - K, 71)])
            batch = np.ascontiguousarray(np.concatenate([blobs, crops]).astype(np.float32))
            g[f"hi_batch_{tag}"] = batch
            g[f"hi_Z_{tag}"] = z.transform(batch).data
            H, W = K + 9, K + 14
            crop = np.ascontiguousarray(lattice[30:30 + H, 50:50 + W])
            g[f"hi_frame_{tag}"] = crop
            Zf = z.transform(crop.astype(np.float64)).data
            ri, ci = sample_index(H, 8), sample_index(W, 9)
            g[f"hi_Zf_{tag}"] = Zf[:, ri][:, :, ci]
            g[f"hi_Zf_sum_{tag}"] = Zf.sum(axis=(1, 2))
            g[f"hi_Zf_max_{tag}"] = np.abs(Zf).max()
            print(tag, batch.shape, g[f"hi_Z_{tag}"].shape, g[f"hi_Zf_{tag}"].shape)
        # the reference's own tail (zmoments.rot_maps / |to_complex| / mirror_map, _zmoments.py:300-316, 420-493) of its dense
        # moments where the build takes them from the matrix-core kernel: n_max 20 (fused band kernel) and 28 (planes tail)
        for n_max, K in ((20, 40), (28, 56)):
            tag = f"{n_max}_{K}"
            crop = np.ascontiguousarray(lattice[90:90 + K + 7, 20:20 + K + 12])
            g[f"hi_maps_frame_{tag}"] = crop
            zm = ZPs(n_max, K).transform(crop.astype(np.float64))
            ri, ci = sample_index(crop.shape[0], 6), sample_index(crop.shape[1], 7)
            g[f"hi_maps_rot_{tag}"] = zm.rot_maps([2, 3, 4, 6])[:, ri][:, :, ci]
            g[f"hi_maps_abs_{tag}"] = np.abs(zm.to_complex().data)[:, ri][:, :, ci]
            g[f"hi_maps_mirror_{tag}"] = zm.mirror_map()[ri][:, ci]
            print("maps", tag, g[f"hi_maps_abs_{tag}"].shape)
        # moments at key points at an order the matrix-core kernel serves: the reference's KeyPoints (features/_keypoint.py:53-78:
        # border clearing, windows cut around the rounded points) + its ZPs(20, 40).transform of those windows
        import make_golden_keypoints
        kp, ZPs2, HoneyComb2 = make_golden_keypoints.import_reference()
        rng = np.random.default_rng(20261006)
        frame = HoneyComb2(size=200, l=12, seed=5).to_image()[:180]                   # (180, 200) float32
        pts = np.column_stack([rng.uniform(-5, 205, 260), rng.uniform(-5, 185, 260)])
        k = kp.KeyPoints(pts, frame, 40)
        g["hi_kp_frame"], g["hi_kp_pts"], g["hi_kp_kept_40"] = frame, pts, k.pts
        g["hi_kp_Z_20_40"] = ZPs2(20, 40).transform(k.extract_patches()).data
        print("key points", g["hi_kp_kept_40"].shape, g["hi_kp_Z_20_40"].shape)
    np.savez(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB,", len(g), "arrays")


if __name__ == "__main__":
    main()
