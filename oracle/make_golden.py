#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE implementation (build container only).

TEST INFRASTRUCTURE.  This script imports the two hot-path modules of jiadongdan/motif-learn
straight from /root/reference (``mtflearn/features/_zps.py``, ``_zmoments.py``) plus its
dataset helpers, under stub packages so that the package ``__init__`` files (which need
scikit-image / numba, absent here) never execute -- SURVEY 8c.  It records inputs and the
reference's outputs as plain arrays.  No reference source or bytecode is copied anywhere; the
fixtures are data.  /root/reference does not exist on the GPU box, so nothing at test/bench
time may import this file's dependencies: tests read only the .npz files.

Usage:  python oracle/make_golden.py   (writes tests/golden/zps_golden.npz)
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "zps_golden.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("mtflearn", "mtflearn.features", "mtflearn.datasets"):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = mod
    from mtflearn.features._zps import ZPs
    from mtflearn.features import _zmoments
    from mtflearn.datasets._zps_test_data import get_zps_test_patches
    from mtflearn.datasets._honeycomb_lattice import HoneyCombLattice
    return ZPs, _zmoments, get_zps_test_patches, HoneyCombLattice


STRUCTURED_ORDERS = ((10, 32), (12, 64), (14, 32), (16, 32), (18, 40), (20, 40), (22, 48), (24, 48))


def sample_index(n, step):
    """Every `step`-th index plus the last one (the zero-padded far border)."""
    return np.array(sorted(set(range(0, n, step)) | {n - 1}), dtype=np.int64)


def structured(g, ZPs, get_patches, HoneyComb):
    lattice = HoneyComb(size=256, l=12, seed=7).to_image()                      # float32 (256, 256)
    for n_max, K in STRUCTURED_ORDERS:
        z = ZPs(n_max, K)
        tag = f"{n_max}_{K}"
        # batch path (_zps.py:146-157): the reference's own test blobs (3- and 6-fold) + windows of a honeycomb lattice
        blobs = np.concatenate([get_patches(size=K, n_fold=3, num_patches=5), get_patches(size=K, n_fold=6, num_patches=3)])
        rows = range(3, 256 - K, 61)
        cols = range(5, 256 - K, 47)
        crops = np.array([lattice[r:r + K, c:c + K] for r in rows for c in cols])
        batch = np.ascontiguousarray(np.concatenate([blobs, crops]).astype(np.float32))
        g[f"st_batch_{tag}"] = batch
        g[f"st_Z_{tag}"] = z.transform(batch).data
        # dense path (_zps.py:159-193) on a float64-cast crop (the FFT oracle is then exact to ~1e-16 max|Z|): every moment
        # plane at a strided set of positions that includes both zero-padded borders, plus per-plane sums over ALL positions
        H, W = K + 11, K + 18
        crop = np.ascontiguousarray(lattice[40:40 + H, 60:60 + W])
        g[f"st_frame_{tag}"] = crop
        Zf = z.transform(crop.astype(np.float64)).data
        ri, ci = sample_index(H, 4), sample_index(W, 5)
        g[f"st_Zf_{tag}"] = Zf[:, ri][:, :, ci]
        g[f"st_Zf_sum_{tag}"] = Zf.sum(axis=(1, 2))
        g[f"st_Zf_max_{tag}"] = np.abs(Zf).max()
    # ---- configs[4]'s tail on a structured crop: the reference's rot_maps / |to_complex| / mirror_map (_zmoments.py:300-316, 420-493)
    # of its own dense moments at n_max 10, 32-px windows -- what the fused device kernel replaces end to end
    z10 = ZPs(10, 32)
    crop = np.ascontiguousarray(lattice[100:100 + 46, 30:30 + 58])                   # (46, 58) float32
    g["st_maps_frame_10_32"] = crop
    zm = z10.transform(crop.astype(np.float64))
    ri, ci = sample_index(46, 3), sample_index(58, 4)
    g["st_maps_rot_10_32"] = zm.rot_maps([2, 3, 4, 6])[:, ri][:, :, ci]
    g["st_maps_abs_10_32"] = np.abs(zm.to_complex().data)[:, ri][:, :, ci]
    g["st_maps_mirror_10_32"] = zm.mirror_map()[ri][:, ci]
    g["st_maps_rot_pnone_unsel012_10_32"] = zm.rot_maps([3, 5], p=None, m_unselect=(0, 1, 2))[:, ri][:, :, ci]
    # ---- configs[0]: the reference's own test image (datasets/_zps_test_data.py:62-65 -- HoneyCombLattice(size=512, l=12),
    # seeded here), 32-px patches, n_max 8: batch path on a strided grid of windows, dense path on the whole frame
    frame = HoneyComb(size=512, l=12, seed=0).to_image()
    assert frame.dtype == np.float32 and frame.shape == (512, 512)
    g["c0_frame_512"] = frame
    z8 = ZPs(8, 32)
    grid = np.arange(0, 512 - 32 + 1, 16)
    g["c0_grid"] = grid
    win = np.array([frame[r:r + 32, c:c + 32] for r in grid for c in grid])
    g["c0_Z_grid_8_32"] = z8.transform(win).data                                   # (961, 45)
    ri = sample_index(512, 13)
    g["c0_sample_index"] = ri
    Zd = z8.transform(frame.astype(np.float64)).data                               # (45, 512, 512), exact FFT oracle
    g["c0_Zf_f64cast_sample"] = Zd[:, ri][:, :, ri]
    g["c0_Zf_f64cast_sum"] = Zd.sum(axis=(1, 2))
    g["c0_Zf_f64cast_abs_sum"] = np.abs(Zd).sum(axis=(1, 2))
    g["c0_Zf_f64cast_max"] = np.abs(Zd).max()
    Zs = z8.transform(frame).data                                                  # float32 image: single-precision FFT inside
    g["c0_Zf_f32_sample"] = Zs[:, ri][:, :, ri]
    g["c0_Zf_f32_sum"] = Zs.sum(axis=(1, 2))


def main():
    ZPs, zm, get_patches, HoneyComb = import_reference()
    g = {}
    rng = np.random.default_rng(20261003)

    # ---- bases ------------------------------------------------------------------------
    z8 = ZPs(8, 32)
    g["basis_8_32"] = z8.polynomials
    g["n_8"], g["m_8"] = z8.n, z8.m
    z59 = ZPs(5, 9)
    g["basis_5_9"] = z59.polynomials
    z1011 = ZPs(10, 11)          # odd size with pixels at rho == 1 up to rounding (6-8-10 triple)
    g["basis_10_11"] = z1011.polynomials
    z48 = ZPs(4, 8)
    g["basis_4_8"] = z48.polynomials
    z1264 = ZPs(12, 64)
    g["basis_12_64_sample"] = z1264.polynomials[:, ::7, ::5]
    g["basis_12_64_sums"] = z1264.polynomials.reshape(91, -1).sum(axis=1)
    g["basis_12_64_abs_sums"] = np.abs(z1264.polynomials).reshape(91, -1).sum(axis=1)
    z1032 = ZPs(10, 32)

    # ---- batch path (_zps.py:146-157) ----------------------------------------------------
    blobs = get_patches(size=32, n_fold=3, num_patches=4)      # float32, the reference's own generator
    g["blobs_32"] = blobs
    g["Z_blobs_8_32"] = z8.fit_transform(blobs).data
    p32 = rng.random((70, 32, 32), dtype=np.float32)           # 70: one full wave + a ragged tail
    g["rand_f32_70_32"] = p32
    g["Z_rand_f32_8_32"] = z8.transform(p32).data
    g["Z_rand_f32_10_32"] = z1032.transform(p32).data
    p64 = rng.standard_normal((5, 9, 9))
    g["rand_f64_5_9"] = p64
    g["Z_rand_f64_5_9"] = z59.transform(p64).data
    p11 = rng.random((6, 11, 11), dtype=np.float32)
    g["rand_f32_6_11"] = p11
    g["Z_rand_f32_10_11"] = z1011.transform(p11).data
    pk64 = rng.random((3, 64, 64), dtype=np.float32)
    g["rand_f32_3_64"] = pk64
    g["Z_rand_f32_12_64"] = z1264.transform(pk64).data
    pint = rng.integers(0, 255, size=(3, 8, 8)).astype(np.uint8)
    g["rand_u8_3_8"] = pint
    g["Z_rand_u8_4_8"] = z48.transform(pint).data

    # ---- dense path (_zps.py:159-193) -- float64 image so the FFT oracle is exact to 1e-16 ----
    frame = HoneyComb(size=96, l=12, seed=0).to_image()[20:68, 10:66]   # (48, 56) float32
    g["frame_f32_48_56"] = frame
    g["Zf_frame_f64cast_8_32"] = z8.transform(frame.astype(np.float64)).data
    g["Zf_frame_f32_8_32_maxabs"] = np.abs(z8.transform(frame).data).max()
    small = rng.standard_normal((20, 23))
    g["frame_f64_20_23"] = small
    g["Zf_small_5_9"] = z59.transform(small).data
    g["Zf_small_4_8"] = z48.transform(small).data
    g["Zf_small_10_11"] = z1011.transform(small).data

    # ---- container post-processing (_zmoments.py:238-493) -----------------------------------
    zb = z8.transform(blobs)
    zc = zb.to_complex()
    g["pp2_complex"], g["pp2_complex_n"], g["pp2_complex_m"] = zc.data, zc.n, zc.m
    g["pp2_rot_maps"] = zb.rot_maps([2, 3, 4, 6])
    g["pp2_rot_maps_pnone"] = zb.rot_maps([3, 6], p=None)
    g["pp2_rot_maps_unsel012"] = zb.rot_maps([3], m_unselect=(0, 1, 2))
    g["pp2_mirror"] = zb.mirror_map()
    g["pp2_norm2"] = zb.normalize(order=2).data
    g["pp2_norm_none"] = zb.normalize().data
    r30 = zb.rotate(30.0)
    g["pp2_rotate30"] = r30.data
    back = zc.to_real()
    g["pp2_toreal"], g["pp2_toreal_n"], g["pp2_toreal_m"] = back.data, back.n, back.m
    sel = zb.select([1, -2])
    g["pp2_select"], g["pp2_select_n"], g["pp2_select_m"] = sel.data, sel.n, sel.m
    uns = zb.unselect([0, 1])
    g["pp2_unselect_m"] = uns.m

    z612 = ZPs(6, 12)
    fsmall = HoneyComb(size=64, l=6, seed=1).to_image()[5:29, 7:35]       # (24, 28)
    g["frame_f32_24_28"] = fsmall
    zf = z612.transform(fsmall.astype(np.float64))
    g["pp3_moments"] = zf.data
    g["pp3_valid_mask"] = zf.valid_mask
    zfc = zf.to_complex()
    g["pp3_complex"] = zfc.data
    g["pp3_rot_maps"] = zf.rot_maps([2, 3, 4, 6])
    g["pp3_mirror"] = zf.mirror_map()
    g["pp3_rotate45"] = zf.rotate(45.0).data
    g["pp3_toreal"] = zfc.to_real().data
    g["pp3_norm2"] = zf.normalize(order=2).data
    g["valid_mask_9_20_23"] = z59.transform(small).valid_mask

    # ---- index algebra ---------------------------------------------------------------------
    g["cmat_8"] = zm.construct_complex_matrix(z8.n, z8.m)
    inv, nr, mr = zm.construct_real_matrix(zc.n, zc.m)
    g["rmat_8"], g["rmat_8_n"], g["rmat_8_m"] = inv, nr, mr
    g["rotmat_8"] = zm.construct_rot_maps_matrix([1, 2, 3, 4, 6], z8.m)
    g["nm2j_8"] = zm.nm2j(z8.n, z8.m)
    g["nm2j_complex_8"] = zm.nm2j_complex(zc.n, zc.m)


    # ---- structured inputs at the orders the reference's own estimator returns (12 .. size / 2,
    # features/_estimate_n_max.py:95,123) -- round 4.  White noise hides an absolute floor (every moment has the
    # same magnitude); on blobs / lattice crops most moments are small against max|Z|, so the elementwise
    # criterion of SURVEY 8c (rtol 1e-6, floor <= 1e-12 max|Z|) is a real test there.
    # (one BLAS thread: the bits of np.dot then do not depend on how many cores the regenerating machine has)
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        structured(g, ZPs, get_patches, HoneyComb)

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **g)
    print(f"wrote {OUT}: {len(g)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
