#!/usr/bin/env python3
"""Generate tests/golden/keypoints_golden.npz from the REFERENCE's key-point code (build container only).

TEST INFRASTRUCTURE.  ``mtflearn/features/_keypoint.py`` imports ``skimage.filters`` at module level (absent here); as in
oracle/make_golden_pickers.py the module is loaded under stub packages with an EMPTY stand-in for ``skimage.filters`` whose
``threshold_li`` / ``threshold_otsu`` raise if called -- they only satisfy the import statement.  Recorded: what never reaches
them -- ``clear_border``, ``KeyPoints.__init__`` / ``extract_patches`` / ``clear_border`` / ``refine``, ``center_of_mass_refine``,
``disk_patch`` -- and the reference's Zernike moments of the extracted patches (``ZPs.transform``, ``_zps.py:146-157``).
``com_refine`` IS the scikit-image thresholds: parity unpinned.  No reference source or bytecode is copied; the fixtures are data.

Usage:  python oracle/make_golden_keypoints.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "keypoints_golden.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("mtflearn", "mtflearn.features", "mtflearn.datasets"):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = mod

    def absent(*a, **k):
        raise RuntimeError("scikit-image is not installed: this stand-in only satisfies the import")
    for name, attrs in (("skimage", ()), ("skimage.filters", ("threshold_li", "threshold_otsu"))):
        mod = types.ModuleType(name)
        for a in attrs:
            setattr(mod, a, absent)
        sys.modules[name] = mod
    from mtflearn.features import _keypoint
    from mtflearn.features._zps import ZPs
    from mtflearn.datasets._honeycomb_lattice import HoneyCombLattice
    return _keypoint, ZPs, HoneyCombLattice


def main():
    kp, ZPs, HoneyComb = import_reference()
    g = {}
    rng = np.random.default_rng(20261005)
    frame = HoneyComb(size=160, l=12, seed=3).to_image()[:150]                   # (150, 160) float32: not square
    g["frame"] = frame
    pts = np.column_stack([rng.uniform(-5, 165, 400), rng.uniform(-5, 155, 400)])   # (x, y), some outside / near the border
    g["pts"] = pts
    for size in (32, 33):
        k = kp.KeyPoints(pts, frame, size)
        g[f"kept_{size}"] = k.pts
        g[f"clear_border_{size}"] = kp.clear_border(pts, frame.shape, size)
        patches = k.extract_patches()
        g[f"patches_{size}"] = patches
        g[f"patches_flat_{size}_head"] = k.extract_patches(flat=True)[:3]
        g[f"Z_{size}"] = ZPs(8, size).transform(patches).data
        k.clear_border(48)                                                            # the method with the shape[1]-for-y quirk
        g[f"kept_after_48_{size}"] = k.pts
    k = kp.KeyPoints(pts, frame, 24)
    g["patches_16_of_24"] = k.extract_patches(16)
    k.refine(r=3)
    g["refined_r3"] = k.pts
    k2 = kp.KeyPoints(pts, frame, 24)
    k2.refine(r=4, mode='disk')
    g["refined_r4_disk"] = k2.pts
    g["disk_5"] = kp.disk_patch(5)
    ipts = np.rint(kp.clear_border(pts, frame.shape, 24)).astype(int)[:40]
    g["com_refine_box"] = kp.center_of_mass_refine(frame, ipts, size=2)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **g)
    print(f"wrote {OUT}: {len(g)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
