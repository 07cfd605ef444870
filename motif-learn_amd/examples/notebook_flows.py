#!/usr/bin/env python3
"""The reference notebooks' two Zernike flows on a synthetic frame, through the drop-in API.

  notebook 2 (batch):  key points -> patches -> ZPs.fit_transform(patches) -> rotation-invariant |Z_nm| -> rot_maps
  notebook 3 (dense):  ZPs.fit_transform(frame) -> rot_maps / mirror_map / |Z_nm| maps

Only the import line differs from the reference (`from mtflearn import ZPs`).  Needs an MI355X.
Run:  python motif-learn_amd/examples/notebook_flows.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mtflearn_amd import ZPs                         # reference: from mtflearn import ZPs
from mtflearn_amd.synthetic import honeycomb_frame


def main():
    frame = honeycomb_frame(1024, seed=7)            # float32 STEM-like frame (the notebooks load a .npy here)
    size, n_max = 32, 10
    zps = ZPs(n_max=n_max, size=size)

    # ---- notebook 2: patches at key points -------------------------------------------------------
    rng = np.random.default_rng(0)
    pts = rng.uniform(size, 1024 - size, size=(5000, 2))                 # stand-in for the peak finder
    ipts = np.rint(pts).astype(int)
    patches = np.array([frame[y - 16:y + 16, x - 16:x + 16] for x, y in ipts])   # KeyPoints.extract_patches
    t = time.perf_counter()
    zm = zps.fit_transform(patches)                                        # (5000, 66) float64, on the GPU
    print(f"batch  : {patches.shape} -> {zm.data.shape} in {1e3 * (time.perf_counter() - t):.1f} ms")
    invariants = np.abs(zm.to_complex().data)                              # rotation-invariant features
    folds = zm.rot_maps([2, 3, 4, 6])                                      # (5000, 4) symmetry scores
    print("         |Z_nm| features", invariants.shape, " rot_maps", folds.shape,
          " dominant fold of patch 0:", [2, 3, 4, 6][int(np.argmax(folds[0]))])
    same = zps.transform_at(frame, pts)                                    # no patch batch at all (extension)
    print("         transform_at == transform(patches):", np.allclose(same.data, zm.data, rtol=1e-9, atol=1e-13))

    # ---- notebook 3: dense symmetry maps -----------------------------------------------------------
    t = time.perf_counter()
    zf = zps.fit_transform(frame)                                          # (66, 1024, 1024) float64
    print(f"dense  : {frame.shape} -> {zf.data.shape} in {1e3 * (time.perf_counter() - t):.1f} ms")
    rot = zf.rot_maps([2, 3, 4, 6])                                        # reference call, NumPy on the host
    t = time.perf_counter()
    maps = zps.symmetry_maps(frame, n_folds=[2, 3, 4, 6])                  # same maps, fused on the GPU
    print(f"fused  : rot_maps + |Z_nm| + mirror_map in {1e3 * (time.perf_counter() - t):.1f} ms;"
          f" max |rot - host rot| = {np.nanmax(np.abs(maps['rot_maps'] - rot)):.2e}")
    valid = zf.valid_mask
    print("         valid (un-padded) positions:", int(valid.sum()), "of", valid.size)


if __name__ == "__main__":
    main()
